#!/bin/bash
# average duration per kernel (and per launch parity for kernels launched twice per chain) of one bench run with
# one context: rocprofv3 kernel trace, alone-on-the-device numbers
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency --contexts 1 --steps 2 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pk/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    d[n].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
tot=0
rows=[]
for k,v in d.items():
    v.sort()
    s=sum(x[1] for x in v); tot+=s
    rows.append((s,k,len(v)))
for s,k,n in sorted(rows,reverse=True)[:24]:
    v=d[k]
    extra=""
    if k in ("replay_search_kernel<true>",):
        ev=[x[1] for x in v[0::2]]; od=[x[1] for x in v[1::2]]
        extra=" even %.1f odd %.1f"%(sum(ev)/len(ev)/1e3,sum(od)/max(1,len(od))/1e3)
    print("%-44s n %5d avg_us %9.1f total_ms %8.2f%s"%(k[:44],n,s/n/1e3,s/1e6,extra))
print("total ms %.2f"%(tot/1e6))
PY
