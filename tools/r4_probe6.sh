#!/bin/bash
# tools/r4_probe6.sh: the forked front end (lpx_set_fork) against the plain one, several context counts and side pools
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p6; mkdir -p $O
DEV=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
run() { tag=$1; shift; env "$@" python3 bench.py --workload ${W:-stream} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 8 --warmup 3 $ARGS 2>$O/$tag.err | tail -1 > $O/$tag.json
  python3 -c "import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['completion']['p50_frame_completion_ms'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'])"; }
ARGS="" run base A=1
ARGS="--fork" run fork20 A=1
ARGS="--fork --contexts 18 --frames-per-step 1152" run fork18 A=1
ARGS="--fork --contexts 16 --frames-per-step 1024" run fork16 A=1
ARGS="--contexts 16 --frames-per-step 1024" run base16 A=1
ARGS="--fork" run fork20_p2 LPX_LIB=$DEV LPX_FORK_STREAMS=2
ARGS="--fork" run fork20_p8 LPX_LIB=$DEV LPX_FORK_STREAMS=8
ARGS="--fork --overlap --contexts 10 --frames-per-step 640" run fork_ovl10 A=1
ARGS="--overlap --contexts 10 --frames-per-step 640" run ovl10 A=1
ARGS="--fork --overlap --contexts 9 --frames-per-step 576" run fork_ovl9 A=1
W=synth1m ARGS="" run s1m A=1
W=synth1m ARGS="" run s1m_grid LPX_LIB=$DEV LPX_CC=grid
