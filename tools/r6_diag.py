"""Round 6 diagnostic: what the LIST path of a large synthetic frame spends (counters, per-group work of the neighbour
kernel, stage times alone).  python tools/r6_diag.py synth5m|synth1m"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from util import synthetic_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "synth5m"
if which == "synth1m":
    pts = synthetic_scene(600_000, 2000, 200, 20240601)
    scfg = SegmentationConfiguration(number_of_planar_partitions=12, number_of_iterations=3)
    ccfg = ClusteringConfiguration(0.09, 0.5)
else:
    pts = synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0)
    scfg = SegmentationConfiguration(number_of_planar_partitions=24, number_of_iterations=3)
    ccfg = ClusteringConfiguration(0.04, 0.5)
ctx = Context(0)
ctx.set_neighbour_mode("lists")
L = ctx._L
G = 1 << 18
L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
for rep in range(3):
    r = ctx.segment_cluster(pts, scfg, ccfg)
st = ctx.frame_stats()
print(which, "N", pts.shape[0], "stats", st, "workspace", ctx.workspace_bytes())
print("entries per obstacle point", st["neighbour_entries"] / max(1, st["n_obstacle"]), "words reserved per point",
      st["neighbour_words"] / max(1, st["n_obstacle"]), "replay entries per expansion",
      st["replay_entries"] / max(1, st["expansions"]), "expansions per point", st["expansions"] / max(1, st["n_obstacle"]))
ctx.profile_enable(True)
for rep in range(5):
    ctx.segment_cluster(pts, scfg, ccfg)
prof = ctx.profile_read()
print("stage ms per frame:", {k: round(v[0] / 5, 3) for k, v in prof.items() if v[1]}, "launches",
      {k: v[1] // 5 for k, v in prof.items() if v[1]})
ctx.profile_enable(False)
assert L.lpx_dbg_group_stats(ctx._h, G, None) == 0
ctx.segment_cluster(pts, scfg, ccfg)
out = np.zeros((G, 8), np.uint32)
assert L.lpx_dbg_group_stats(ctx._h, G, out.ctypes.data_as(C.c_void_p)) == 0
used = out[4096:]
used = used[used[:, 2] > 0]
T, ncur, nq, tot, c1, c2, c0 = [used[:, i].astype(np.float64) for i in range(7)]
print("groups", len(used))
for name, v in [("T", T), ("intervals", ncur), ("queries", nq), ("reserved/query", tot / nq), ("kcycles traverse", c0 / 1e3),
                ("kcycles alloc", c1 / 1e3),
                ("kcycles total", c2 / 1e3)]:
    print(f"  {name:14s} mean {v.mean():9.1f}  p50 {np.percentile(v, 50):9.1f}  p90 {np.percentile(v, 90):9.1f} "
          f" p99 {np.percentile(v, 99):9.1f}  max {v.max():9.1f}")
b = used[:, 2] > 1
print("  bucket groups:", int(b.sum()), "T mean", T[b].mean(), "cycles", c2[b].mean(), " single groups:", int((~b).sum()),
      "T mean", T[~b].mean(), "cycles", c2[~b].mean())
ctx.close()
