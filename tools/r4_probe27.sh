#!/bin/bash
# tools/r4_probe27.sh: the launch-shape knobs on 1M-point chains (they were tuned on 120k-point frames): replay workgroups
# per frame, kd tail staging, kd group size, chain shapes
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p27; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
run() {  # name args env...
  local name=$1 args=$2; shift 2
  env "$@" python3 bench.py --workload synth1m --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 4 --warmup 1 $args 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'])"
}
run base1 "" X=1
run rs8 "" LPX_RS_GRID=8
run rs16 "" LPX_RS_GRID=16
run rs32 "" LPX_RS_GRID=32
run rs2 "" LPX_RS_GRID=2
run tail512 "" LPX_KD_TAIL=512
run tail1984 "" LPX_KD_TAIL=1984
run bucket32 "" LPX_IX_BUCKET=32
run bucket64 "" LPX_IX_BUCKET=64
run base2 "" X=1
run c8b16 "--contexts 8 --batch 16 --frames-per-step 128" X=1
run c16b16 "--contexts 16 --batch 16 --frames-per-step 256" X=1
run c12b32 "--contexts 12 --batch 32 --frames-per-step 384" X=1
run c6b32 "--contexts 6 --batch 32 --frames-per-step 192" X=1
run c4b64 "--contexts 4 --batch 64 --frames-per-step 256" X=1
run base3 "" X=1
