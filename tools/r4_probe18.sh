#!/bin/bash
# tools/r4_probe18.sh: 5M-point frames with twelve in flight -- where the top kd levels hand over from the
# four-launch rounds (kd_top_*) to one workgroup per range (LPX_KD_HAND, LPX_KD_TOP_MIN; development build)
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p18; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
B="--workload synth5m --no-cpu-baseline --no-latency --no-inflight --no-sub"
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $B --steps 3 --warmup 1 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'])"
}
run base X=1
run hand128k LPX_KD_HAND=131072
run hand512k LPX_KD_HAND=524288
run hand2m LPX_KD_HAND=2097152
run min512k LPX_KD_TOP_MIN=524288
run min512k_hand256k LPX_KD_TOP_MIN=524288 LPX_KD_HAND=262144
run notop LPX_KD_TOP_MIN=100000000
run base2 X=1
