#!/bin/bash
# tools/r4_probe20.sh: the memory REQUESTS of every kernel of one chain (L2 <-> fabric: TCC_EA0_RDREQ / WRREQ; CU -> L2:
# TCP_TCC_*; L2 hits and misses), with the line-request burner of r4_probe14 in the same run as the yardstick
# (LPX_BURN_MEM=96: 25.2 M scattered 64-byte lines per launch)
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p20; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so LPX_BURN_MEM=96
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify"
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pm_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$name -o e -- python3 $GRAFT_REPO_ROOT/bench.py $B --contexts 1 --frames-per-step 64 --steps 2 --warmup 1 > $O/pmc_$name.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_$name > $O/pmc_$name.json 2>>$O/pmc_$name.log
  echo "pmc $name: $(wc -c < $O/pmc_$name.json) bytes"
}
pmc ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
pmc ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum
pmc l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
pmc lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
