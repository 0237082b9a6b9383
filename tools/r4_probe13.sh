#!/bin/bash
# tools/r4_probe13.sh: the x gather inside the last sort pass and plane pass 0 inside the seed selection (two launches fewer per chain)
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
REPS="1" bash tools/ab_run.sh r4ab13 stream "ps2 default ps2 default" --steps 8 --warmup 3 --contexts 16 --frames-per-step 1024
REPS="1" bash tools/ab_run.sh r4ab13 synth1m "ps2 default" --steps 4 --warmup 1
REPS="1" bash tools/ab_run.sh r4ab13 synth5m "ps2 default" --steps 4 --warmup 1
