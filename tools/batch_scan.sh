#!/bin/bash
# throughput of bench.py over (frames per launch chain, concurrent contexts); diagnostic
for cfg in "32 8 256 2 16" "32 8 256 1 16" "32 8 256 4 16" "32 8 256 2 8" "32 8 256 2 32" "32 10 320 2 16" "32 6 192 2 16" "48 6 288 2 16" "64 4 256 2 16"; do
  set -- $cfg
  echo -n "batch=$1 contexts=$2 frames/step=$3 threads=$4 hwq=$5: "
  GPU_MAX_HW_QUEUES=$5 python bench.py --steps 6 --warmup 2 --batch $1 --contexts $2 --frames-per-step $3 --threads $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mpts/s', d['ms_per_step'], 'ms/step')"
done
