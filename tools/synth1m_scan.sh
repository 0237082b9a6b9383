run() { python3 bench.py --workload synth1m --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 3 --warmup 1 $1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('synth1m [$1]', d['value'], d['ms_per_step'])"; }
run "--contexts 8 --batch 16 --frames-per-step 128"
run "--overlap --contexts 4 --batch 16 --frames-per-step 64"
run "--overlap --contexts 8 --batch 16 --frames-per-step 128"
run "--overlap --contexts 8 --batch 8 --frames-per-step 64"
run "--contexts 12 --batch 16 --frames-per-step 192"
run "--contexts 8 --batch 32 --frames-per-step 256"
