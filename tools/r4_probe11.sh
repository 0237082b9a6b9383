#!/bin/bash
# tools/r4_probe11.sh: point-pair scan width of the cell linking (same box), and what the memory pipeline does under load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p11; mkdir -p $O
REPS="1" bash tools/ab_run.sh r4ab11 stream "ps1 ps2 ps3 default ps1 ps2" --steps 8 --warmup 3 --contexts 16 --frames-per-step 1024
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
for C in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/p6
  LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/ab/liblpx_ps1.so rocprofv3 --pmc $C --output-format csv -d /tmp/p6 -o f -- python3 $GRAFT_REPO_ROOT/bench.py --workload stream --contexts 16 --frames-per-step 1024 --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/pmc_$tag.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p6 > $GRAFT_REPO_ROOT/$O/pmc_$tag.json
done
