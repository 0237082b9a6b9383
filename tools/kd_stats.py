import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from lidar_processing_amd import Context
from util import load_frame
pts = load_frame("0000000000")
obs = pts[oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6, number_of_iterations=5))["obstacle_idx"]]
ctx = Context(0); ctx.reserve(obs.shape[0])
L = ctx._L
L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
assert L.lpx_dbg_group_stats(ctx._h, 4096, None) == 0
ctx.dbg_kd_layout(obs)
out = np.zeros((4096, 8), np.uint32)
assert L.lpx_dbg_group_stats(ctx._h, 4096, out.ctypes.data_as(C.c_void_p)) == 0
d = out.ravel()[:64]
print("n", d[31], "total kcycles", d[30] / 1e3)
prev = 0; prevr = 0
for s in range(12):
    if d[2 * s] == 0: break
    print(f"  sub-level {s}: {(d[2*s]-prev)/1e3:8.1f} kcycles  rounds {d[2*s+1]-prevr}")
    prev = d[2 * s]; prevr = d[2*s+1]
print("  leaf:", (d[30] - prev) / 1e3, "kcycles")
