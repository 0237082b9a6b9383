#!/bin/bash
# throughput of bench.py over (frames per launch chain, concurrent contexts); diagnostic
for cfg in "1 32 64 4" "4 16 64 4" "8 8 128 2" "16 4 128 2" "16 8 128 2" "32 4 128 2" "32 2 128 1" "64 2 128 1"; do
  set -- $cfg
  echo -n "batch=$1 contexts=$2 frames/step=$3 threads=$4: "
  python bench.py --steps 6 --warmup 2 --batch $1 --contexts $2 --frames-per-step $3 --threads $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mpts/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
done
