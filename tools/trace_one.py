"""Per-dispatch durations of one frame from a rocprofv3 kernel trace (diagnostic)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the ingest kernel marks the last frame
starts = [i for i, r in enumerate(rows) if "ingest_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:38]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s start %8.1f us  dur %8.1f us  gap %6.1f  grid %s wg %s lds %s" % (
        n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"]))
    prev_end = e
