#!/bin/bash
# tools/r4_probe12.sh: the replay with two expansions in flight (variant builds rspair8 / rspair4) against the single pipeline, same box
cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/lidar_processing_amd/ab
for L in rspair8 rspair4; do
  LPX_LIB=$AB/liblpx_$L.so python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_batch.py -m gpu -x -q -k "stream_154 or test_batch or bench_shape" 2>&1 | tail -2
done
REPS="1" bash tools/ab_run.sh r4ab12 stream "default rspair8 rspair4 rs4 default rspair4" --steps 8 --warmup 3 --contexts 16 --frames-per-step 1024
REPS="1" bash tools/ab_run.sh r4ab12 synth1m "default rspair8 rspair4" --steps 4 --warmup 1
