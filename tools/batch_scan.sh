#!/bin/bash
# throughput of bench.py over (frames per launch chain, concurrent contexts); diagnostic
for cfg in "16 12 192 2" "16 16 256 2" "12 16 192 2" "24 10 240 2" "32 8 256 2" "32 12 384 2" "16 12 192 4" "20 12 240 2"; do
  set -- $cfg
  echo -n "batch=$1 contexts=$2 frames/step=$3 threads=$4: "
  python bench.py --steps 6 --warmup 2 --batch $1 --contexts $2 --frames-per-step $3 --threads $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mpts/s', d['ms_per_step'], 'ms/step')"
done
