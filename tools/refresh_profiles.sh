#!/bin/bash
# Regenerates the measured artefacts under gpurun_out/<tag>/ on the GPU box (copy them to profiles/ afterwards):
# per workload the bench line, the rocprofv3 kernel stats of the default command and of --contexts 1 (launch
# durations of the kernels alone), and FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes).
TAG=${1:-r04}
shift
WORKLOADS=${@:-kitti synth1m synth5m stream}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  case $W in kitti) S="--steps 10 --warmup 3";; stream) S="--steps 6 --warmup 2";; *) S="--steps 4 --warmup 1";; esac
  # counter passes: one context; the 5M-point workload with two frames per step (rocprofv3 --pmc crashed at start-up
  # with eight 5M-point frames resident)
  case $W in synth5m) PF="--frames-per-step 2";; synth1m) PF="--frames-per-step 8";; *) PF="--frames-per-step 64";; esac
  python3 $GRAFT_REPO_ROOT/bench.py --workload $W $S 2>$O/${TAG}_${W}_bench.err | tail -1 > $O/${TAG}_${W}_bench.json
  rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o a -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W $S --no-cpu-baseline --no-latency --no-inflight --no-sub > $O/rocprof_${W}_default.log 2>&1
  cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_default_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W $S --contexts 1 --no-cpu-baseline --no-latency --no-inflight --no-sub > $O/rocprof_${W}_contexts1.log 2>&1
  cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_contexts1_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -o c -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --contexts 1 $PF --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 2 --warmup 1 > $O/pmc_${W}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -o d -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --contexts 1 $PF --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 2 --warmup 1 > $O/pmc_${W}_write.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p3 /tmp/p4 > $O/${TAG}_${W}_pmc_fetch_write_per_kernel.json
  echo "== $W"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/${TAG}_${W}_contexts1_kernel_stats.csv 8
  cut -c1-200 $O/${TAG}_${W}_bench.json
done
