run() { env $1 python3 bench.py --workload $3 --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 6 --warmup 2 --contexts $2 --batch $4 --frames-per-step $(( $2 * $4 )) $5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1] $3 contexts $2 x $4 $5', d['value'], d['ms_per_step'])"; }
run "A=1" 10 stream 128
run "A=1" 10 stream 96
run "A=1" 20 stream 96 --no-overlap
run "A=1" 16 stream 128 --no-overlap
run "A=1" 20 stream 128 --no-overlap
