#!/bin/bash
# tools/pmc_run.sh OUT WORKLOAD [bench args]: FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, one context)
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -o c -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --contexts 1 --frames-per-step ${PMC_FRAMES:-64} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 2 --warmup 1 "$@" > $O/pmc_${W}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -o d -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --contexts 1 --frames-per-step ${PMC_FRAMES:-64} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 2 --warmup 1 "$@" > $O/pmc_${W}_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p3 /tmp/p4 > $O/${W}_pmc_fetch_write_per_kernel.json
python3 - <<PY
import json
d=json.load(open("$O/${W}_pmc_fetch_write_per_kernel.json"))
rows=[]; tot=0
for k,v in d.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or k.startswith("copy_kernel") or "fillBuffer" in k or "at::" in k: continue
    n=v["FETCH_SIZE"]["launches"]; per=(2*v["FETCH_SIZE"]["avg"]+v["WRITE_SIZE"]["avg"])/1024
    rows.append((per*n,k[:46],per,n)); tot+=per*n
chains=d["frame_init_kernel"]["FETCH_SIZE"]["launches"]
for t,k,per,n in sorted(rows,reverse=True)[:26]: print("%-46s %8.1f MB/launch x %5.1f per chain = %8.1f MB per chain"%(k,per,n/chains,t/chains))
print("total per chain %.1f MB, per frame %.1f MB (%d frames per chain)"%(tot/chains, tot/chains/${PMC_FRAMES:-64}, ${PMC_FRAMES:-64}))
PY
