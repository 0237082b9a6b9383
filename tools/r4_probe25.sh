#!/bin/bash
# tools/r4_probe25.sh: 1M-point chains -- the tile-sum scan of the label scan with 256 threads (variant new against head),
# and the kd levels' 1024-thread workgroups (LPX_KD_WIDE)
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p25; mkdir -p $O
run() {  # name lib env...
  local name=$1 lib=$2; shift 2
  env LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/ab/liblpx_$lib.so "$@" python3 bench.py --workload synth1m --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 4 --warmup 1 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'])"
}
run head1 head X=1
run new1 new X=1
run head2 head X=1
run new2 new X=1
run wide2m new LPX_KD_WIDE=2000000
run wide300k new LPX_KD_WIDE=300000
run wide2m_b new LPX_KD_WIDE=2000000
run new3 new X=1
