for st in 10 60 10; do
python3 bench.py --workload kitti --no-cpu-baseline --no-latency --no-inflight --no-sub --steps $st --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kitti steps $st', d['value'], d['ms_per_step'])"
done
python3 bench.py --workload stream --no-cpu-baseline --no-latency --no-inflight --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream', d['value'], d['kitti_3_frames_cycled'])"
