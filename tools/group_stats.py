"""Distribution of per-group work in the neighbour kernel (diagnostic)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from lidar_processing_amd import Context, ClusteringConfiguration  # noqa: E402
from util import load_frame  # noqa: E402

for frame in ["0000000000", "0000000153"]:
    pts = load_frame(frame)
    obs = pts[oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6, number_of_iterations=5))["obstacle_idx"]]
    ctx = Context(0)
    ctx.reserve(obs.shape[0])
    G = 4096
    L = ctx._L
    L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    assert L.lpx_dbg_group_stats(ctx._h, G, None) == 0
    ctx.cluster(obs, ClusteringConfiguration(0.25, 0.5))
    out = np.zeros((G, 8), np.uint32)
    assert L.lpx_dbg_group_stats(ctx._h, G, out.ctypes.data_as(C.c_void_p)) == 0
    used = out[out[:, 2] > 0]
    T, ncur, nq, hits, c1, c2 = [used[:, i].astype(np.float64) for i in range(6)]
    print(frame, "groups", len(used), "M", obs.shape[0])
    for name, v in [("T", T), ("intervals", ncur), ("hits/query", hits / nq), ("kcycles alloc", c1 / 1e3),
                    ("kcycles total", c2 / 1e3)]:
        print(f"  {name:14s} mean {v.mean():9.1f}  p50 {np.percentile(v, 50):9.1f}  p90 {np.percentile(v, 90):9.1f} "
              f" p99 {np.percentile(v, 99):9.1f}  max {v.max():9.1f}")
    b = used[:, 2] > 1
    print("  bucket groups: T mean", T[b].mean(), " single groups: T mean", T[~b].mean())
    ctx.close()
