"""How much of a bench step is host enqueue time?  (diagnostic)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame
dev = torch.device("cuda", 0)
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5)
ccfg = ClusteringConfiguration(0.25, 0.5)
frames = [load_frame(f) for f in FRAMES]
recs = []
for hf in frames:
    rec = np.zeros((hf.shape[0], 8), np.float32); rec[:, :4] = hf
    recs.append(torch.from_numpy(rec).to(dev))
nmax = max(f.shape[0] for f in frames)
C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 64
ctxs = [Context(0) for _ in range(C)]
for c in ctxs: c.reserve(nmax)
outs = [dict(l=torch.empty(nmax, dtype=torch.int32, device=dev), g=torch.empty(nmax, dtype=torch.int32, device=dev),
             o=torch.empty(nmax, dtype=torch.int32, device=dev), p=torch.empty(24, dtype=torch.float32, device=dev),
             c=torch.empty(nmax, dtype=torch.int32, device=dev), n=torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(F)]
def step():
    for j in range(F):
        fid = j % 3; o = outs[j]
        ctxs[j % C].segment_cluster_device(recs[fid].data_ptr(), 32, frames[fid].shape[0], scfg, ccfg, o["l"].data_ptr(),
                                           o["g"].data_ptr(), o["o"].data_ptr(), o["p"].data_ptr(), o["c"].data_ptr(), o["n"].data_ptr())
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); te = 0.0
K = 5
for _ in range(K):
    a = time.perf_counter(); step(); te += time.perf_counter() - a
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"contexts {C}: total {tot/K*1e3:.2f} ms/step, host enqueue {te/K*1e3:.2f} ms/step ({te/K/F*1e6:.1f} us per frame call)")
