"""Per-component statistics of the replay kernel (diagnostic)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from lidar_processing_amd import ClusteringConfiguration, Context
from util import load_frame
for frame in ["0000000000", "0000000153"]:
    pts = load_frame(frame)
    obs = pts[oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6, number_of_iterations=5))["obstacle_idx"]]
    ctx = Context(0); ctx.reserve(obs.shape[0])
    L = ctx._L
    L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    ctx.cluster(obs, ClusteringConfiguration(0.25, 0.5))
    assert L.lpx_dbg_group_stats(ctx._h, 8192, None) == 0
    ctx.cluster(obs, ClusteringConfiguration(0.25, 0.5))
    out = np.zeros((8192, 8), np.uint32)
    assert L.lpx_dbg_group_stats(ctx._h, 8192, out.ctypes.data_as(C.c_void_p)) == 0
    d = out[:4096]
    d = d[d[:, 0] > 0]
    order = np.argsort(-d[:, 5].astype(np.int64))
    print(frame, "components", len(d), "total expansions", d[:, 1].sum(), "total kcycles", d[:, 5].sum() / 1e3)
    print("   members expansions entries windows seeds kcycles  cyc/exp  entries/exp  pf_hits")
    for i in order[:6]:
        m, x, e, w, sd, c, b = [int(v) for v in d[i, :7]]
        print(f"   {m:7d} {x:10d} {e:7d} {w:7d} {sd:5d} {c/1e3:8.1f} {c/max(x,1):8.0f} {e/max(x,1):10.1f} {b:6d}")
    ctx.close()
