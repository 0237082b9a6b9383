#!/bin/bash
# tools/r4_probe24.sh: seed selection spread over many workgroups for long segments (selw_*) against the (segment, z)
# sort -- its tests, the whole suite, then the synthetic workloads with LPX_SEEDS=sort (the path before) and without
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 1500 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q -k "wide_selection or long_segments or seed_representatives or single_workgroup_limit or full_size" 2>&1 | tail -4
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
O=$GRAFT_REPO_ROOT/gpurun_out/r4p24; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
run() {  # name workload env...
  local name=$1 w=$2; shift 2
  env "$@" python3 bench.py --workload $w --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 4 --warmup 1 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['verified']['mismatches'], {k:round(v,4) for k,v in d['roofline']['stage_ms_per_launch_alone'].items() if v and k in ('xsort','gather','zsort','seeds')})"
}
run s1m_sort synth1m LPX_SEEDS=sort
run s1m_wide synth1m X=1
run s1m_sort2 synth1m LPX_SEEDS=sort
run s1m_wide2 synth1m X=1
run s5m_sort synth5m LPX_SEEDS=sort
run s5m_wide synth5m X=1
run s5m_sort2 synth5m LPX_SEEDS=sort
run s5m_wide2 synth5m X=1
