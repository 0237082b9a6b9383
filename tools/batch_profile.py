"""Per-stage HIP-event time of ONE launch chain over B frames, nothing else on the GPU (diagnostic):
shows which stages scale with the number of frames (they fill the device) and which stay flat."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from util import FRAMES, load_frame  # noqa: E402

SEG = dict(number_of_planar_partitions=6, number_of_iterations=5)
CLU = dict(distance_squared=0.25, cluster_quality=0.5)


def run(B, reps=5):
    dev = torch.device("cuda:0")
    frames = [load_frame(f) for f in FRAMES]
    pitch = max(f.shape[0] for f in frames)
    host = np.zeros((B, pitch, 8), np.float32)
    n = np.zeros(B, np.uint32)
    for b in range(B):
        f = frames[b % len(frames)]
        host[b, :f.shape[0], :4] = f
        n[b] = f.shape[0]
    d_pts = torch.from_numpy(host).to(dev)
    lab = torch.empty((B, pitch), dtype=torch.int32, device=dev)
    gi = torch.empty((B, pitch), dtype=torch.int32, device=dev)
    oi = torch.empty((B, pitch), dtype=torch.int32, device=dev)
    pl = torch.empty((B, 24), dtype=torch.float32, device=dev)
    cl = torch.empty((B, pitch), dtype=torch.int32, device=dev)
    cnt = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    ctx = Context(0, batch=B)
    ctx.reserve(pitch)
    scfg, ccfg = SegmentationConfiguration(**SEG), ClusteringConfiguration(**CLU)

    def go():
        ctx.segment_cluster_batch_device(n, d_pts.data_ptr(), 32, pitch, scfg, ccfg, lab.data_ptr(), gi.data_ptr(),
                                         oi.data_ptr(), pl.data_ptr(), cl.data_ptr(), cnt.data_ptr())
    go()
    ctx.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile_enable(True)
    for _ in range(reps):
        go()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    tot = sum(v[0] for v in prof.values()) / reps
    print(f"== B={B}: chain {wall:.3f} ms = {wall / B:.4f} ms/frame = {n.sum() / wall / 1e3:.1f} Mpts/s; "
          f"sum of stages {tot:.3f} ms")
    return {k: v[0] / reps for k, v in prof.items()}


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [1, 4, 16]
    res = {B: run(B) for B in Bs}
    print("%-14s" % "stage" + "".join("%12s" % f"B={B}" for B in Bs) + "   (ms per launch group)")
    for k in res[Bs[0]]:
        print("%-14s" % k + "".join("%12.4f" % res[B][k] for B in Bs))
