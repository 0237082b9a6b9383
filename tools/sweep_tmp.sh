run() { name=$1; shift
  v=$(env "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$name $v"; }
B="python3 bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 2"
run "kitti default" $B
run "kitti ctx1" $B --contexts 1
run "kitti ctx1 rsgrid8" LPX_RS_GRID=8 $B --contexts 1
run "kitti ctx4" $B --contexts 4 --frames-per-step 128
run "kitti ctx4 rsgrid8" LPX_RS_GRID=8 $B --contexts 4 --frames-per-step 128
run "stream" $B --workload stream
run "synth1m" $B --workload synth1m
python3 tools/stage_latency.py 0000000077 30 2>&1 | grep wall
