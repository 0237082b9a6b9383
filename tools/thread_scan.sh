#!/bin/bash
for t in 1 2 4 8; do
  echo -n "threads=$t  "
  python bench.py --threads $t --no-cpu-baseline --steps 6 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mpts/s", d["config"]["frames_per_s"], "fps", d["roofline"]["kernel"])'
done
