#!/bin/bash
# tools/shape_scan.sh WORKLOAD "C:B C:B ..." : throughput for contexts x frames-per-chain shapes (512 frames per step)
W=$1; shift
for cb in $1; do
  C=${cb%%:*}; B=${cb##*:}
  python3 bench.py --workload $W --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 6 --warmup 2 --contexts $C --batch $B --frames-per-step $((C*B)) 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W contexts $C batch $B', d['value'], d['ms_per_step'])"
done
