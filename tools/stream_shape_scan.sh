for cfg in "15 462" "20 616" "20 640" "16 512" "22 704"; do set -- $cfg
python3 bench.py --workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 6 --warmup 2 --contexts $1 --frames-per-step $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream contexts $1 frames $2', d['value'], d['ms_per_step'])"
done
