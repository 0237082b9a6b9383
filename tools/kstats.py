"""Prints a rocprofv3 *_kernel_stats.csv compactly."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print("%-44s calls %6d total_ms %9.2f avg_us %9.1f %6.2f%%" % (
        n[:44], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
        float(r["Percentage"])))
print("sum of kernel durations: %.2f ms" % (tot / 1e6))
