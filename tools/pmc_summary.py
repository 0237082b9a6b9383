"""Summarises rocprofv3 --pmc counter_collection csv files: average counter value per kernel launch."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in sorted(out.items()):
    res[k] = {cn: {"launches": len(v), "avg": sum(v) / len(v)} for cn, v in c.items()}
print(json.dumps(res, indent=1))
