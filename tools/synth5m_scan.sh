run() { python3 bench.py --workload synth5m --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 3 --warmup 1 $1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('synth5m [$1]', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_launch_alone'])"; }
run "--contexts 12 --frames-per-step 12"
run "--search --batch 4 --contexts 4 --frames-per-step 16"
run "--search --batch 8 --contexts 2 --frames-per-step 16"
run "--search --batch 1 --contexts 12 --frames-per-step 12"
