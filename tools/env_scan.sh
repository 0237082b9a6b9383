#!/bin/bash
# tools/env_scan.sh WORKLOAD "VAR=val VAR2=val;VAR=val2;..." [bench args]: one short bench line per environment setting
W=$1; SETS=$2; shift 2
IFS=';' read -ra LIST <<< "$SETS"
for s in "${LIST[@]}"; do
  env $s python3 bench.py --workload $W --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W [$s]', d['value'], d['ms_per_step'], 'replay alone', d['roofline']['stage_ms_per_launch_alone'].get('replay'))"
done
