"""Where a sequencer wavefront of replay_search_kernel spends its cycles (needs the LPX_RS_PROF variant build:
tools/build_variant.sh rsprof -DLPX_RS_PROF; LPX_LIB=lidar_processing_amd/ab/liblpx_rsprof.so python tools/replay_prof.py)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame, synthetic_scene

which = sys.argv[1] if len(sys.argv) > 1 else "kitti"
if which == "kitti":
    frames = [load_frame(f) for f in FRAMES]; B = 32
    scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
else:
    frames = [synthetic_scene(600_000, 2000, 200, 20240601)]; B = 8
    scfg = SegmentationConfiguration(number_of_planar_partitions=12, number_of_iterations=3); ccfg = ClusteringConfiguration(0.09, 0.5)
pitch = max(f.shape[0] for f in frames)
host = np.zeros((B, pitch, 8), np.float32)
n = np.zeros(B, np.uint32)
for b in range(B):
    f = frames[b % len(frames)]; host[b, :f.shape[0], :4] = f; n[b] = f.shape[0]
dev = torch.device("cuda", 0)
d_pts = torch.from_numpy(host).to(dev)
outs = [torch.empty((B, pitch), dtype=torch.int32, device=dev) for _ in range(4)]
d_planes = torch.empty((B, 4 * scfg.number_of_planar_partitions), dtype=torch.float32, device=dev)
d_counts = torch.zeros((B, 4), dtype=torch.int32, device=dev)
ctx = Context(0, batch=B); ctx.set_neighbour_mode("search"); ctx.reserve(pitch)
L = ctx._L
L.lpx_dbg_group_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
def run():
    ctx.segment_cluster_batch_device(n, d_pts.data_ptr(), 32, pitch, scfg, ccfg, outs[0].data_ptr(), outs[1].data_ptr(),
                                     outs[2].data_ptr(), d_planes.data_ptr(), outs[3].data_ptr(), d_counts.data_ptr())
    ctx.synchronize()
run(); run()
NG = 16384  # 512 KiB: the upper half holds the sequencer records
assert L.lpx_dbg_group_stats(ctx._h, NG, None) == 0
run()
raw = np.zeros(NG * 4, np.uint64)
assert L.lpx_dbg_group_stats(ctx._h, NG, raw.ctypes.data_as(C.c_void_p)) == 0
raw = raw[NG * 2:]
nrec = int(min(raw[0], 4000))
rr = raw[8:8 + 8 * nrec].reshape(nrec, 8)
r = rr.astype(np.float64)
names = ["total", "seed scan", "window set-up", "table+cull+issue", "candidates", "apply"]
tot = r[:, 0].sum()
exp = r[:, 6].sum()
win = float((rr[:, 7] >> np.uint64(32)).sum())
ent = float((rr[:, 7] & np.uint64(0xffffffff)).sum())
print(f"{which}: {nrec} sequencer wavefronts, {int(exp)} expansions, {int(win)} windows, {int(ent)} hits; cycles per expansion "
      f"{tot / max(exp, 1):.0f} (sum over all wavefronts)")
for k in range(1, 6):
    print(f"  {names[k]:18s} {100 * r[:, k].sum() / tot:5.1f} %   {r[:, k].sum() / max(exp, 1):8.0f} cycles per expansion")
print(f"  other              {100 * (tot - r[:, 1:6].sum()) / tot:5.1f} %")
i = int(np.argmax(r[:, 0]))
print(f"slowest wavefront: {r[i, 0] / 1e6:.2f} Mcycles, {int(r[i, 6])} expansions, {r[i, 0] / max(r[i, 6], 1):.0f} cycles per expansion:",
      ", ".join(f"{names[k]} {100 * r[i, k] / r[i, 0]:.0f}%" for k in range(1, 6)))
ctx.close()
