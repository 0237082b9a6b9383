#!/bin/bash
# tools/r4_probe9.sh: register-staged point-pair scans in the cell linking, two kd levels per round trip in the chunk-table traversal
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p9; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --workload ${W:-stream} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 8 --warmup 3 $ARGS 2>$O/$tag.err | tail -1 > $O/$tag.json
  python3 -c "import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['completion']['p50_frame_completion_ms'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'], {k:round(v,3) for k,v in d['roofline']['stage_ms_per_launch_alone'].items() if v})"; }
ARGS="--contexts 16 --frames-per-step 1024" run c16 A=1
ARGS="--contexts 18 --frames-per-step 1152" run c18 A=1
ARGS="" run c20 A=1
W=synth1m ARGS="" run s1m A=1
W=synth5m ARGS="" run s5m A=1
