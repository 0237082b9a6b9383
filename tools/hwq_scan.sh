#!/bin/bash
# throughput vs number of HIP hardware queues (diagnostic)
for q in 4 8 16 32; do
  echo -n "GPU_MAX_HW_QUEUES=$q  "
  GPU_MAX_HW_QUEUES=$q python bench.py --contexts 32 --frames-per-step 64 --no-cpu-baseline --steps 6 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mpts/s", d["config"]["frames_per_s"], "fps")'
done
