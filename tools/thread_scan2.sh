for t in 2 4 6 10; do
python3 bench.py --workload kitti --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 10 --warmup 3 --threads $t 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kitti threads $t', d['value'], d['ms_per_step'])"
done
