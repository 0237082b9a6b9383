#!/bin/bash
# Regenerates the measured artefacts under gpurun_out/<tag>/ on the GPU box (copy them to profiles/ afterwards):
# per workload the bench line, the rocprofv3 kernel stats of the default command (under load) and of --contexts 1 (the
# kernels alone), FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes); for the stream and configs[2] also the memory REQUESTS per
# chain (tools/probe.sh requests) and the vector-ALU activity (SQ_ACTIVE_INST_VALU); and the hash of the library's
# sources (lpx_build_info) that bench.py compares with the loaded library's (`traffic_stale`).
# usage: tools/refresh_profiles.sh TAG [workloads...]
TAG=${1:-r05}
shift
WORKLOADS=${@:-kitti synth1m synth5m stream}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 - > $O/${TAG}_source_hash.json <<PY
import json, sys
sys.path.insert(0, "$R")
from lidar_processing_amd import _lib
info = _lib.lib().lpx_build_info().decode()
print(json.dumps({"library_source_hash": info.rsplit("src ", 1)[1].strip() if "src " in info else None,
                  "build_info": info, "made_by": "tools/refresh_profiles.sh $TAG $WORKLOADS"}))
PY
cat $O/${TAG}_source_hash.json
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  case $W in kitti) S="--steps 10 --warmup 3";; stream) S="--steps 6 --warmup 2";; *) S="--steps 4 --warmup 1";; esac
  python3 $R/bench.py --workload $W $S 2>$O/${TAG}_${W}_bench.err | tail -1 > $O/${TAG}_${W}_bench.json
  rm -rf /tmp/p1 /tmp/p2
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o a -- python3 $R/bench.py --workload $W $S --no-cpu-baseline --no-latency --no-inflight --no-sub --no-dist-selftest > $O/rocprof_${W}_default.log 2>&1
  cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_default_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b -- python3 $R/bench.py --workload $W $S --contexts 1 --no-cpu-baseline --no-latency --no-inflight --no-sub --no-dist-selftest > $O/rocprof_${W}_contexts1.log 2>&1
  cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_contexts1_kernel_stats.csv
  # counter passes: one context, one chain per step (synth5m: two frames, rocprofv3 --pmc crashed with eight resident)
  (cd $R && tools/probe.sh traffic $TAG $W > $O/traffic_$W.log 2>&1)
  cp $O/${W}_pmc_fetch_write_per_kernel.json $O/${TAG}_${W}_pmc_fetch_write_per_kernel.json
  if [ $W = stream ] || [ $W = synth1m ] || [ $W = synth5m ]; then
    (cd $R && tools/probe.sh requests $TAG $W > $O/requests_$W.log 2>&1)
    python3 - $O/requests_$W.json > $O/${TAG}_${W}_requests.json <<'PY'
import json, sys
t = open(sys.argv[1]).read()
print(json.dumps(json.loads(t[:t.rindex('\n{"fabric')]), indent=1))
PY
    (cd $R && tools/probe.sh pmc $TAG $W active "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" >> $O/requests_$W.log 2>&1)
    cp $O/pmc_active.json $O/${TAG}_${W}_pmc_active.json
  fi
  echo "== $W"; python3 $R/tools/kstats.py $O/${TAG}_${W}_contexts1_kernel_stats.csv 8
  cut -c1-200 $O/${TAG}_${W}_bench.json
done
