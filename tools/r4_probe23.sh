#!/bin/bash
# tools/r4_probe23.sh: is a request that HITS a cache as scarce as one that goes to memory?  The line-request burner of
# r4_probe14 with its lines drawn from 4 GB (memory), 64 MB (last-level cache) and 2 MB (one L2)
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p23; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify"
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $B --steps 6 --warmup 2 --contexts 16 --frames-per-step 1024 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"
}
run base1 X=1
run mem96_4g LPX_BURN_MEM=96
run mem96_64m LPX_BURN_MEM=96 LPX_BURN_SPAN_MB=64
run mem96_2m LPX_BURN_MEM=96 LPX_BURN_SPAN_MB=2
run mem384_2m LPX_BURN_MEM=384 LPX_BURN_SPAN_MB=2
run mem384_64m LPX_BURN_MEM=384 LPX_BURN_SPAN_MB=64
run base2 X=1
cd /tmp && export TMPDIR=/tmp
for S in 4096 64 2; do
  rm -rf /tmp/pb$S
  LPX_BURN_MEM=384 LPX_BURN_SPAN_MB=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$S -o b -- python3 $GRAFT_REPO_ROOT/bench.py $B --contexts 1 --frames-per-step 64 --steps 2 --warmup 1 > $O/alone$S.log 2>&1
  echo "alone span $S MB, 384 loads per thread: $(grep burn_mem $(find /tmp/pb$S -name '*kernel_stats.csv' | head -1) | awk -F, '{print $(NF-4)}') ns"
done
