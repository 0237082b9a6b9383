#!/bin/bash
# tools/r4_probe1.sh: round-4 baseline on today's box + what each clustering stage costs the device under load
# (LPX_SKIP drops a stage: results are wrong, only the rates are read)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p1; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --workload ${W:-stream} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 8 --warmup 3 $ARGS 2>$O/$tag.err | tail -1 > $O/$tag.json
  python3 -c "import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'])"; }
ARGS="" run base A=1
ARGS="" run skip_replay LPX_SKIP=replay
ARGS="" run skip_grid LPX_SKIP=grid,replay
ARGS="" run skip_kd_index LPX_SKIP=kd,index,replay
ARGS="" run skip_sort LPX_SKIP=sort,replay
ARGS="" run skip_all_cluster LPX_SKIP=kd,index,grid,sort,replay
ARGS="--overlap --contexts 10 --frames-per-step 640" run ovl10 A=1
ARGS="--overlap --contexts 11 --frames-per-step 704" run ovl11 A=1
ARGS="--overlap --contexts 10 --frames-per-step 640" run ovl10_skip_replay LPX_SKIP=replay
ARGS="--contexts 22 --frames-per-step 1408" run c22 A=1
W=synth1m ARGS="" run s1m A=1
W=synth5m ARGS="" run s5m A=1
