"""kernel_trace.csv of rocprofv3 -> stream, short kernel name, start, end (ns) as a small gzip'd csv"""
import csv, gzip, re, sys
rd = csv.DictReader(open(sys.argv[1]))
with gzip.open(sys.argv[2], "wt") as out:
    for r in rd:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        out.write("%s;%s;%s;%s;%s\n" % (r["Queue_Id"], r["Stream_Id"], name, r["Start_Timestamp"], r["End_Timestamp"]))
