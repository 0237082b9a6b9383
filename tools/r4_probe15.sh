#!/bin/bash
# tools/r4_probe15.sh: components from the sweep over y-sorted x slabs (LPX_CC=sweep) against the clique-cell grid --
# parity first, then the headline shape and the kernels alone, same box, development library for both
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p15; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "component_searches" 2>&1 | tail -5
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub"
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $B --steps 6 --warmup 2 --contexts 16 --frames-per-step 1024 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'], d['verified']['mismatches'], {k:round(v,4) for k,v in d['roofline']['stage_ms_per_launch_alone'].items() if v})"
}
run grid1 LPX_CC=grid
run plain1 LPX_CC=sweep LPX_SWEEP_PLAIN=1
run sweep1 LPX_CC=sweep
run grid2 LPX_CC=grid
run sweep2 LPX_CC=sweep
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb && LPX_CC=sweep rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-verify --contexts 1 --frames-per-step 64 --steps 3 --warmup 1 > $O/sweep_alone.log 2>&1; cp $(find /tmp/pb -name '*kernel_stats.csv' | head -1) $O/sweep_alone_kernel_stats.csv)
head -30 $O/sweep_alone_kernel_stats.csv | cut -c1-150
