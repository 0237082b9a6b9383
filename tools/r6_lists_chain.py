"""Round 6: what a chain of B frames in LIST mode costs call by call on a fresh batch context (the frames-in-flight
curve of the stream line showed 64 frames in flight in list mode at 75 Mpts/s with a p99 of 3.1 s: the workspace that now
starts small grows on the evidence of the frames it sees, and this prints when, by how much and what each call cost).
usage: r6_lists_chain.py [B] [chains] [contexts]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from util import load_stream_frame, stream_names  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 12
C = int(sys.argv[3]) if len(sys.argv) > 3 else 2
frames = [load_stream_frame(nm) for nm in stream_names()]
F = len(frames)
pitch = max(f.shape[0] for f in frames) + 7
host = np.zeros((F, pitch, 4), np.float32)
for i, f in enumerate(frames):
    host[i, :f.shape[0]] = f[:, :4]
dev = torch.device("cuda:0")
d_pts = torch.from_numpy(host).to(dev)
n = np.array([f.shape[0] for f in frames], np.uint32)
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5)
ccfg = ClusteringConfiguration(0.25, 0.5)


def run(i, out):
    torch.cuda.set_device(0)
    ctx = Context(0, batch=B)
    ctx.set_neighbour_mode("lists")
    ctx.reserve(pitch)
    bufs = [torch.zeros((B, pitch), dtype=torch.int32, device=dev) for _ in range(4)]
    planes = torch.zeros((B, 24), dtype=torch.float32, device=dev)
    counts = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    k = i * B
    for c in range(CH):
        lo = k % (F - B + 1)
        a = time.perf_counter()
        ctx.segment_cluster_batch_device(n[lo:lo + B], d_pts[lo].data_ptr(), 16, pitch, scfg, ccfg, bufs[0].data_ptr(),
                                         bufs[1].data_ptr(), bufs[2].data_ptr(), planes.data_ptr(), bufs[3].data_ptr(),
                                         counts.data_ptr())
        e = time.perf_counter()
        ctx.synchronize()
        b = time.perf_counter()
        st = counts.cpu().numpy()[:, 3]
        out.append((i, c, lo, round((e - a) * 1e3, 2), round((b - a) * 1e3, 2), ctx.workspace_bytes(), int((st != 0).sum())))
        k += C * B
    ctx.close()


outs = [[] for _ in range(C)]
th = [threading.Thread(target=run, args=(i, outs[i])) for i in range(C)]
for t in th:
    t.start()
for t in th:
    t.join()
for o in outs:
    for r in o:
        print("ctx %d chain %2d frames %3d..  enqueue %9.2f ms  complete %9.2f ms  workspace %s  bad status %d" % r)
