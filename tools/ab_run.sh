#!/bin/bash
# tools/ab_run.sh OUT WORKLOAD "LIB1 LIB2 ..." [bench args]: the bench line of every build (A/B on one box)
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2; LIBS=$3; shift 3
mkdir -p $O
for L in $LIBS; do
  if [ "$L" = default ]; then unset LPX_LIB; else export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/ab/liblpx_$L.so; fi
  for rep in ${REPS:-1 2}; do
    python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-latency --no-inflight --no-sub "$@" 2>$O/${W}_${L}_$rep.err | tail -1 > $O/${W}_${L}_$rep.json
    python3 -c "import json,sys; d=json.load(open('$O/${W}_${L}_$rep.json')); print('$W $L $rep', d['value'], d['ms_per_step'], {k:round(v,4) for k,v in d['roofline']['stage_ms_per_launch_alone'].items() if v})"
  done
done
