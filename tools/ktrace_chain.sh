#!/bin/bash
# kernel timeline of ONE launch chain (one batch context alone on the device): every launch of the last chain with
# its start offset, duration and the idle gap before it.  usage: ktrace_chain.sh [bench args, e.g. --workload kitti]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pkc
rocprofv3 --kernel-trace --output-format csv -d /tmp/pkc -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency --no-inflight --no-sub --contexts 1 --frames-per-step 32 --steps 2 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pkc/**/*kernel_trace.csv",recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,r["Grid_Size_X"],r["Grid_Size_Z"],r["Workgroup_Size_X"],r["LDS_Block_Size"]))
rows.sort()
starts=[i for i,r in enumerate(rows) if r[2].startswith("frame_init_kernel")]
# the third chain from the end is a timed, un-profiled step (the stage profile adds event records, not kernels)
i0=starts[-3]; i1=starts[-2]
t0=rows[i0][0]; prev=t0; busy=0
for s,e,n,gx,gz,wx,lds in rows[i0:i1]:
    print("%9.1f us  dur %8.1f  gap %6.1f  %-44s grid %s x %s wg %s lds %s"%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,n[:44],gx,gz,wx,lds))
    busy+=e-s; prev=e
print("chain span %.1f us, kernels busy %.1f us, launches %d"%((prev-t0)/1e3,busy/1e3,i1-i0))
PY
