"""Round 6 diagnostic: per-component work of the LIST replay of a large synthetic frame (members, expansions, entries,
windows, seeds, cycles): is the kernel's duration one component's serial chain or the sum over the sequencers?"""
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
G = 1 << 18  # (the neighbour kernel writes a row per kd group into the same buffer, unchecked)
L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
for rep in range(2):
    ctx.segment_cluster(pts, scfg, ccfg)
assert L.lpx_dbg_group_stats(ctx._h, G, None) == 0
ctx.segment_cluster(pts, scfg, ccfg)
out = np.zeros((G, 8), np.uint32)
assert L.lpx_dbg_group_stats(ctx._h, G, out.ctypes.data_as(C.c_void_p)) == 0
sel = (out[:65536, 0] > 1) & (out[:65536, 6] == 0)
phases = out[65536:131072][sel].astype(np.float64) * 16.0  # seed search, window set-up, first-list wait, apply
d = out[:65536]
d = d[sel]  # rows the replay wrote (it zeroes word 6; the neighbour kernel's hold cycles there)
m, x, e, w, sd, c = [d[:, i].astype(np.float64) for i in range(6)]
print(which, "components with a sequencer", len(d), "expansions", int(x.sum()), "entries", int(e.sum()), "kcycles summed",
      c.sum() / 1e3)
order = np.argsort(-c)
print("   members expansions entries windows seeds kcycles  cyc/exp  entries/exp")
for i in order[:12]:
    print(f"   {int(m[i]):7d} {int(x[i]):10d} {int(e[i]):7d} {int(w[i]):7d} {int(sd[i]):5d} {c[i]/1e3:8.1f} {c[i]/max(x[i],1):8.0f} "
          f"{e[i]/max(x[i],1):10.1f}")
big = order[:32]
tot = c[big].sum()
print("  32 largest components, share of their cycles: seed search %.1f %%, window set-up %.1f %%, first-list wait %.1f %%, "
      "apply %.1f %%" % tuple(100.0 * phases[big, i].sum() / tot for i in range(4)))
for name, v in [("members", m), ("expansions", x), ("kcycles", c / 1e3), ("cycles/expansion", c / np.maximum(x, 1))]:
    print(f"  {name:18s} mean {v.mean():9.1f}  p50 {np.percentile(v, 50):9.1f}  p90 {np.percentile(v, 90):9.1f} "
          f" p99 {np.percentile(v, 99):9.1f}  max {v.max():9.1f}")
ctx.close()
