#!/bin/bash
# tools/probe.sh -- ONE parameterised measurement script (rounds 1-4 grew 22 one-off launchers; their recipes are in
# docs/experiments.md).  Runs on the GPU box from the repository root; everything lands under gpurun_out/OUT/.
#
#   probe.sh ab      OUT WORKLOAD "LIB LIB ..." [bench args]   bench line of every library, alternating, REPS (default "1 2")
#                                                             rounds; LIB = default | dev | NAME (lidar_processing_amd/ab/liblpx_NAME.so)
#   probe.sh env     OUT WORKLOAD "A=1 B=2;C=3;..." [args]     development library, one bench line per environment setting
#   probe.sh shapes  OUT WORKLOAD "C:B C:B ..." [args]         contexts x frames-per-chain
#   probe.sh trace   OUT WORKLOAD [bench args]                 every launch of ONE chain alone: start, duration, gap (kernel trace)
#   probe.sh stats   OUT WORKLOAD [bench args]                 rocprofv3 --kernel-trace --stats of the command (csv copied)
#   probe.sh pmc     OUT WORKLOAD NAME "COUNTER ..." [args]    one --pmc pass (one context, one chain per step), per-kernel averages
#   probe.sh requests OUT WORKLOAD [args]                      the five request passes: fabric reads / writes + atomics, L2, CU->L2, latency
#   probe.sh traffic OUT WORKLOAD [args]                       FETCH_SIZE / WRITE_SIZE passes
# Environment: LPX_LIB is set per library by `ab`; STEPS / WARMUP (default 6 / 2) for the short lines.
set -u
cmd=$1; OUT=$2; W=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$OUT; mkdir -p $O
ulimit -c 0
SHORT="--workload $W --no-cpu-baseline --no-latency --no-inflight --no-sub --no-dist-selftest --steps ${STEPS:-6} --warmup ${WARMUP:-2}"
ONE="--workload $W --no-cpu-baseline --no-latency --no-inflight --no-sub --no-dist-selftest --no-verify --contexts 1 --steps 2 --warmup 1"
case $W in synth5m) ONE="$ONE --frames-per-step ${PF:-2}";; synth1m) ONE="$ONE --frames-per-step ${PF:-8}";; *) ONE="$ONE --frames-per-step ${PF:-64}";; esac
lib_of() { case $1 in default) echo "";; dev) echo $R/lidar_processing_amd/liblpx_dev.so;; *) echo $R/lidar_processing_amd/ab/liblpx_$1.so;; esac; }
line() {  # tag: prints value, ms per step, p99 and the stage times alone of $O/tag.json
  python3 - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "NO LINE", e); sys.exit(0)
st = {k: round(v, 3) for k, v in d["roofline"]["stage_ms_per_launch_alone"].items() if v}
print(sys.argv[2], d["value"], "Mpts/s", d["ms_per_step"], "ms/step p99", d["completion"]["p99_frame_completion_ms"],
      "mismatches", (d.get("verified") or {}).get("mismatches"), "alone", st)
PY
}
pmc_pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pm_$name
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$name -o e -- python3 $R/bench.py $ONE $EXTRA > $O/pmc_$name.log 2>&1)
  python3 $R/tools/pmc_summary.py /tmp/pm_$name > $O/pmc_$name.json 2>>$O/pmc_$name.log
  echo "pmc $name: $(wc -c < $O/pmc_$name.json) bytes"
}
case $cmd in
ab)
  LIBS=$1; shift
  for rep in ${REPS:-1 2}; do for L in $LIBS; do
    p=$(lib_of $L); if [ -z "$p" ]; then unset LPX_LIB; else export LPX_LIB=$p; fi
    python3 $R/bench.py $SHORT "$@" 2>$O/${W}_${L}_$rep.err | tail -1 > $O/${W}_${L}_$rep.json; line ${W}_${L}_$rep
  done; done;;
env)
  SETS=$1; shift; export LPX_LIB=$(lib_of dev); i=0
  IFS=';' read -ra LIST <<< "$SETS"
  for s in "${LIST[@]}"; do i=$((i+1)); t=${W}_env$i
    env $s python3 $R/bench.py $SHORT "$@" 2>$O/$t.err | tail -1 > $O/$t.json; echo "[$s]"; line $t
  done;;
shapes)
  for cb in $1; do C=${cb%%:*}; B=${cb##*:}; t=${W}_c${C}_b$B
    python3 $R/bench.py $SHORT --contexts $C --batch $B --frames-per-step $((C*B)) "${@:2}" 2>$O/$t.err | tail -1 > $O/$t.json; line $t
  done;;
trace)
  rm -rf /tmp/pkc
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/pkc -o a -- python3 $R/bench.py $ONE "$@" > $O/trace_$W.log 2>&1)
  python3 - $(find /tmp/pkc -name "*kernel_trace.csv" | head -1) > $O/trace_$W.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Grid_Size_X"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("frame_init_kernel")]
i0, i1 = starts[-3], starts[-2]  # a timed, un-profiled chain
t0 = rows[i0][0]; prev = t0; busy = 0
for s, e, n, gx, gz, wx, lds in rows[i0:i1]:
    print("%9.1f us  dur %8.1f  gap %6.1f  %-46s grid %s x %s wg %s lds %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:46], gx, gz, wx, lds))
    busy += e - s; prev = e
print("chain span %.1f us, kernels busy %.1f us, launches %d" % ((prev - t0) / 1e3, busy / 1e3, i1 - i0))
PY
  tail -1 $O/trace_$W.txt;;
stats)
  rm -rf /tmp/pks
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pks -o a -- python3 $R/bench.py $SHORT "$@" > $O/stats_$W.log 2>&1)
  cp $(find /tmp/pks -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv; python3 $R/tools/kstats.py $O/${W}_kernel_stats.csv 12;;
pmc)
  NAME=$1; CTRS=$2; shift 2; EXTRA="$*"; pmc_pass $NAME $CTRS;;
requests)
  EXTRA="$*"
  pmc_pass ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
  pmc_pass ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum
  pmc_pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
  python3 $R/tools/requests_table.py $O/pmc_ea_rd.json $O/pmc_ea_wr.json $O/pmc_l2.json > $O/requests_$W.json; tail -4 $O/requests_$W.json;;
traffic)
  EXTRA="$*"
  pmc_pass fetch FETCH_SIZE; pmc_pass write WRITE_SIZE
  python3 $R/tools/pmc_summary.py /tmp/pm_fetch /tmp/pm_write > $O/${W}_pmc_fetch_write_per_kernel.json;;
*) echo "unknown command $cmd"; exit 2;;
esac
