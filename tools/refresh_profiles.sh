#!/bin/bash
# Regenerates the measured artefacts under gpurun_out/<tag>/ on the GPU box (copy them to profiles/ afterwards):
# bench lines, rocprofv3 kernel stats of the default command and of --contexts 1, FETCH_SIZE / WRITE_SIZE per kernel.
TAG=${1:-r01_g}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json
python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_contexts1.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --no-cpu-baseline > $O/rocprof_contexts1.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_contexts1_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -o c -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -o d -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p3 /tmp/p4 > $O/${TAG}_pmc_fetch_write_per_kernel.json
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/${TAG}_bench_contexts1_kernel_stats.csv 8
cut -c1-160 $O/${TAG}_bench.json
