#!/bin/bash
# tools/r4_probe7.sh: launch-count reductions on the stream, chains of 128 frames (variant library b128), context counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p7; mkdir -p $O
B128=$GRAFT_REPO_ROOT/lidar_processing_amd/ab/liblpx_b128.so
run() { tag=$1; shift; env "$@" python3 bench.py --workload ${W:-stream} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 8 --warmup 3 $ARGS 2>$O/$tag.err | tail -1 > $O/$tag.json
  python3 -c "import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['completion']['p50_frame_completion_ms'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'])"; }
ARGS="" run c20 A=1
ARGS="--contexts 16 --frames-per-step 1024" run c16 A=1
ARGS="--contexts 18 --frames-per-step 1152" run c18 A=1
ARGS="--batch 128 --contexts 8 --frames-per-step 1024" run b128c8 LPX_LIB=$B128
ARGS="--batch 128 --contexts 10 --frames-per-step 1280" run b128c10 LPX_LIB=$B128
ARGS="--batch 128 --contexts 6 --frames-per-step 768" run b128c6 LPX_LIB=$B128
ARGS="--batch 96 --contexts 10 --frames-per-step 960" run b96c10 LPX_LIB=$B128
ARGS="--batch 96 --contexts 12 --frames-per-step 1152" run b96c12 LPX_LIB=$B128
W=synth5m ARGS="--search --batch 2 --contexts 4 --frames-per-step 16" run s5m_search LPX_LIB=$B128
W=synth5m ARGS="" run s5m A=1
