#!/bin/bash
# tools/wave_cycles.sh OUT [bench args]: what every kernel of ONE chain holds of the device -- SQ_WAVE_CYCLES (wave
# residency: the currency of DESIGN.md section 5, point 3), SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY per kernel
# launch (rocprofv3 --pmc, one context, one 64-frame chain per step), sorted by residency.
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p5
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p5 -o e -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${W:-stream} --contexts 1 --frames-per-step ${PMC_FRAMES:-64} --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify --steps 2 --warmup 1 "$@" > $O/pmc_wave.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/p5 > $O/wave_cycles_per_kernel.json
python3 - <<PY
import json
d=json.load(open("$O/wave_cycles_per_kernel.json"))
chains=d["frame_init_kernel"]["SQ_WAVE_CYCLES"]["launches"]
rows=[]
for k,v in d.items():
    if "SQ_WAVE_CYCLES" not in v or k.startswith("copy_kernel") or "at::" in k or "fillBuffer" in k: continue
    n=v["SQ_WAVE_CYCLES"]["launches"]; w=v["SQ_WAVE_CYCLES"]["avg"]
    rows.append((w*n/chains,k[:44],n/chains,w,v.get("SQ_WAIT_ANY",{}).get("avg",0)/max(w,1),v.get("SQ_ACTIVE_INST_ANY",{}).get("avg",0)/max(w,1),v.get("SQ_BUSY_CYCLES",{}).get("avg",0)))
tot=sum(r[0] for r in rows)
print("kernel                                        launches/chain  wave-quadcycles/chain   share  wait  active  busy/launch")
for t,k,n,w,wa,ac,b in sorted(rows,reverse=True)[:30]: print("%-44s %6.1f %16.0f %6.1f%% %5.2f %5.2f %12.0f"%(k,n,t,100*t/tot,wa,ac,b))
print("total wave quad-cycles per chain %.3g"%tot)
PY
