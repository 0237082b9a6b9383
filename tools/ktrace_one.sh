#!/bin/bash
# kernel timeline of ONE frame on a single-frame context (list path): every launch of the last frame with its
# start offset, duration and the idle gap before it.  usage: ktrace_one.sh [frame-id | synth1m | synth5m]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pk1 -o a -- python3 $GRAFT_REPO_ROOT/tools/stage_latency.py ${1:-0000000077} 3 lists > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pk1/**/*kernel_trace.csv",recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n))
rows.sort()
# the last frame: from the last ingest_kernel on (the profiled repetitions add event records, not kernels)
starts=[i for i,r in enumerate(rows) if r[2].startswith("ingest_kernel")]
i0=starts[-1]
t0=rows[i0][0]; prev=t0; busy=0
for s,e,n in rows[i0:]:
    print("%9.1f us  dur %8.1f  gap %6.1f  %s"%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,n[:60]))
    busy+=e-s; prev=e
print("frame span %.1f us, kernels busy %.1f us, launches %d"%((prev-t0)/1e3,busy/1e3,len(rows)-i0))
PY
