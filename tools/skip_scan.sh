for s in none kd index grid sort replay "kd,index,grid,sort,replay"; do
  LPX_SKIP=$s python3 bench.py --workload kitti --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip=$s', d['value'], d['ms_per_step'])"
done
