#!/bin/bash
# tools/r4_probe16.sh: where the time of sweep_link_kernel goes -- the kernel cut short after each of its phases
# (LPX_SWEEP_STOP, development build, results wrong), one 64-frame chain at a time
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p16; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so LPX_CC=sweep
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify"
cd /tmp && export TMPDIR=/tmp
for S in 1 2 3 4 0; do
  rm -rf /tmp/pb$S
  LPX_SWEEP_STOP=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$S -o b -- python3 $GRAFT_REPO_ROOT/bench.py $B --contexts 1 --frames-per-step 64 --steps 3 --warmup 1 > $O/stop$S.log 2>&1
  echo "stop $S: $(grep sweep_link $(find /tmp/pb$S -name '*kernel_stats.csv' | head -1) | awk -F, '{print $(NF-4)}')"
done
