#!/bin/bash
# tools/r4_probe22.sh: the sorts' first pass makes up its values (positions) instead of reading an array ingest / flatten
# had to write -- parity, then the headline shape against the commit before (variant head), one box
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
cp lidar_processing_amd/liblpx_dev.so lidar_processing_amd/ab/liblpx_new.so
REPS="1" bash tools/ab_run.sh r4ab22 stream "head new head new head new" --steps 8 --warmup 3 --contexts 16 --frames-per-step 1024
REPS="1" bash tools/ab_run.sh r4ab22 synth1m "head new" --steps 4 --warmup 1
