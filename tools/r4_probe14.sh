#!/bin/bash
# tools/r4_probe14.sh: what is the device short of with sixteen chains in flight?  (1) instruction mix per kernel of one
# chain (SQ_INSTS_* / SQ_ACTIVE_INST_*), (2) the headline shape with a known amount of vector-ALU work or of scattered
# line requests added to every chain (LPX_BURN_ALU / LPX_BURN_MEM, development build).
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p14; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify"
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $B --steps 6 --warmup 2 --contexts 16 --frames-per-step 1024 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"
}
run base1 X=1
run alu20k LPX_BURN_ALU=20000
run alu60k LPX_BURN_ALU=60000
run mem32 LPX_BURN_MEM=32
run mem96 LPX_BURN_MEM=96
run base2 X=1
run alu120k LPX_BURN_ALU=120000
run mem192 LPX_BURN_MEM=192
# the burners alone (one context): their launch durations
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb && LPX_BURN_ALU=60000 LPX_BURN_MEM=96 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 $GRAFT_REPO_ROOT/bench.py $B --contexts 1 --frames-per-step 64 --steps 2 --warmup 1 > $O/burn_alone.log 2>&1; cp $(find /tmp/pb -name '*kernel_stats.csv' | head -1) $O/burn_alone_kernel_stats.csv)
grep -i "burn" $O/burn_alone_kernel_stats.csv
# instruction mix, one chain at a time
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters_avail.txt | sort -u > $O/sq_counters.txt
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pm_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$name -o e -- python3 $GRAFT_REPO_ROOT/bench.py $B --contexts 1 --frames-per-step 64 --steps 2 --warmup 1 > $O/pmc_$name.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_$name > $O/pmc_$name.json 2>>$O/pmc_$name.log
  echo "pmc $name: $(wc -c < $O/pmc_$name.json) bytes"
}
unset LPX_LIB
pmc insts1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
pmc insts2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES
pmc active1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pmc active2 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pmc busy SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE
