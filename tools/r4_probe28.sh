#!/bin/bash
# tools/r4_probe28.sh: 5M-point frames on twelve single-frame contexts: 1024- against 256-thread workgroups on the kd
# levels (LPX_KD_SOLO=1 / 0; default: 1024 only while the context is alone on the device)
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p28; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
run() {  # name env...
  local name=$1; shift
  env "$@" python3 bench.py --workload synth5m --no-cpu-baseline --no-inflight --no-sub --steps 3 --warmup 1 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'], d['latency'].get('device_resident_ms'))"
}
run solo1 LPX_KD_SOLO=1
run solo0 LPX_KD_SOLO=0
run auto X=1
run solo1b LPX_KD_SOLO=1
run solo0b LPX_KD_SOLO=0
run autob X=1
