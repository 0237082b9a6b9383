"""Concurrency analysis of a rocprofv3 kernel trace of bench.py (diagnostic).

usage: concurrency.py <kernel_trace.csv> [tail_fraction]
Prints, over the last `tail_fraction` of the trace (default 0.5, i.e. the timed steps): kernels in
flight (time average), busy fraction and dispatch gaps per HW queue, dispatch rate, and the per-kernel
share of the in-queue time.
"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
q = np.array([int(r["Queue_Id"]) for r in rows])
sid = np.array([int(r["Stream_Id"]) for r in rows])
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for r in rows]
t0, t1 = st.min(), en.max()
cut = t1 - int((t1 - t0) * frac)
sel = st >= cut
print(f"trace {(t1 - t0) / 1e6:.1f} ms, analysing last {(t1 - cut) / 1e6:.1f} ms: {sel.sum()} dispatches, "
      f"{len(set(q[sel]))} HW queues, {len(set(sid[sel]))} streams")
span = (t1 - cut)
busy = (en[sel] - st[sel]).sum()
print(f"kernels in flight (time average): {busy / span:.2f};  dispatch rate {sel.sum() / (span / 1e9) / 1e3:.1f} k/s "
      f"= one per {span / sel.sum() / 1e3:.2f} us")
# in-flight histogram
ev = np.concatenate([np.stack([st[sel], np.ones(sel.sum(), np.int64)], 1), np.stack([en[sel], -np.ones(sel.sum(), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
lvl = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
hist = defaultdict(int)
for l, d in zip(lvl[:-1], dt):
    hist[int(l)] += int(d)
tot = sum(hist.values())
print("in-flight level : share of time")
for l in sorted(hist):
    if hist[l] / tot > 0.005:
        print(f"   {l:3d} : {hist[l] / tot * 100:5.1f}%")
# per-stream gaps (a stream is one frame chain at a time)
gaps = []
gap_by_next = defaultdict(list)
for s in set(sid[sel]):
    idx = np.where(sel & (sid == s))[0]
    idx = idx[np.argsort(st[idx])]
    g = st[idx][1:] - en[idx][:-1]
    gaps.append(g)
    for i, gg in zip(idx[1:], g):
        gap_by_next[names[i]].append(gg)
g = np.concatenate(gaps)
print(f"per-stream gap end->next start: median {np.median(g) / 1e3:.2f} us, mean {g.mean() / 1e3:.2f} us, "
      f"p90 {np.percentile(g, 90) / 1e3:.2f} us; sum of gaps / sum of kernel time = {g.sum() / busy:.2f}")
per = defaultdict(lambda: [0, 0])
for i in np.where(sel)[0]:
    per[names[i]][0] += 1
    per[names[i]][1] += en[i] - st[i]
print("kernel                                   calls   avg_us  share   mean gap before (us)")
for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
    gb = np.mean(gap_by_next[n]) / 1e3 if gap_by_next[n] else 0.0
    print(f"   {n[:38]:38s} {c:6d} {d / c / 1e3:8.1f} {d / busy * 100:6.1f}%  {gb:8.2f}")
