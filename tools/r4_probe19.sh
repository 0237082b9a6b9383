#!/bin/bash
# tools/r4_probe19.sh: cell table of 1.25 M slots (any size) against 2 M rounded up to a power of two -- parity of the
# grid path, then the headline shape, same box (variant pow2 = the commit before, tools/build_variant.sh)
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -3
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
cp lidar_processing_amd/liblpx_dev.so lidar_processing_amd/ab/liblpx_new.so
REPS="1" bash tools/ab_run.sh r4ab19 stream "pow2 new pow2 new pow2 new" --steps 8 --warmup 3 --contexts 16 --frames-per-step 1024
