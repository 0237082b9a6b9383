"""One launch chain of 32 KITTI frames on a batch context (search mode), twice; for the printf-instrumented variant
builds (LPX_LIB=lidar_processing_amd/ab/liblpx_kdprof.so python tools/one_chain.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame
frames = [load_frame(f) for f in FRAMES]; B = 32
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
pitch = max(f.shape[0] for f in frames)
host = np.zeros((B, pitch, 8), np.float32); n = np.zeros(B, np.uint32)
for b in range(B):
    f = frames[b % len(frames)]; host[b, :f.shape[0], :4] = f; n[b] = f.shape[0]
dev = torch.device("cuda", 0)
d_pts = torch.from_numpy(host).to(dev)
outs = [torch.empty((B, pitch), dtype=torch.int32, device=dev) for _ in range(4)]
d_planes = torch.empty((B, 24), dtype=torch.float32, device=dev); d_counts = torch.zeros((B, 4), dtype=torch.int32, device=dev)
ctx = Context(0, batch=B); ctx.set_neighbour_mode("search"); ctx.reserve(pitch)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ctx.segment_cluster_batch_device(n, d_pts.data_ptr(), 32, pitch, scfg, ccfg, outs[0].data_ptr(), outs[1].data_ptr(),
                                     outs[2].data_ptr(), d_planes.data_ptr(), outs[3].data_ptr(), d_counts.data_ptr())
    ctx.synchronize()
    print("---- chain", rep, flush=True)
ctx.close()
