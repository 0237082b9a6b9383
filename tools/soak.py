"""Soak test: many concurrent multi-frame chains, every output of every frame checked against the single-frame
path after each round (races in the union-find, the histogram tickets or the workspace allocators would show
up as a differing label).  usage: soak.py [seconds] [batch] [contexts] [overlap]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from util import FRAMES, load_frame  # noqa: E402

SEG = dict(number_of_planar_partitions=6, number_of_iterations=5)
CLU = dict(distance_squared=0.25, cluster_quality=0.5)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    overlap = len(sys.argv) > 4 and sys.argv[4] == "overlap"
    dev = torch.device("cuda:0")
    scfg, ccfg = SegmentationConfiguration(**SEG), ClusteringConfiguration(**CLU)
    frames = [load_frame(f) for f in FRAMES]
    ref_ctx = Context(0)
    refs = [ref_ctx.segment_cluster(f, scfg, ccfg) for f in frames]
    ref_ctx.close()
    pitch = max(f.shape[0] for f in frames)
    F = B * C
    ids = [(3 * j + j // 7) % len(frames) for j in range(F)]
    host = np.zeros((F, pitch, 8), np.float32)
    for j, fid in enumerate(ids):
        host[j, :frames[fid].shape[0], :4] = frames[fid]
    n = np.array([frames[fid].shape[0] for fid in ids], np.uint32)
    d_pts = torch.from_numpy(host).to(dev)
    lab = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    gi = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    oi = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    pl = torch.empty((F, 24), dtype=torch.float32, device=dev)
    cl = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    cnt = torch.zeros((F, 4), dtype=torch.int32, device=dev)
    ctxs = [Context(0, batch=B) for _ in range(C)]
    for c in ctxs:
        c.reserve(pitch)
        if overlap:
            c.set_overlap(True)
    t_end = time.time() + seconds
    rounds = 0
    while time.time() < t_end:
        lab.fill_(-5)
        cl.fill_(-5)
        cnt.zero_()
        torch.cuda.synchronize()
        for rep in range(3):  # back-to-back chains on every context, then one check
            for k, c in enumerate(ctxs):
                lo = k * B
                c.segment_cluster_batch_device(n[lo:lo + B], d_pts[lo].data_ptr(), 32, pitch, scfg, ccfg,
                                               lab[lo].data_ptr(), gi[lo].data_ptr(), oi[lo].data_ptr(),
                                               pl[lo].data_ptr(), cl[lo].data_ptr(), cnt[lo].data_ptr())
        for c in ctxs:
            c.synchronize()
        counts = cnt.cpu().numpy().view(np.uint32)
        h_lab, h_cl, h_oi = lab.cpu().numpy(), cl.cpu().numpy(), oi.cpu().numpy()
        for j, fid in enumerate(ids):
            r = refs[fid]
            ng, no, nc, status = (int(v) for v in counts[j])
            assert status == 0 and nc == r["n_clusters"] and no == len(r["obstacle_idx"]), (rounds, j, counts[j])
            assert np.array_equal(h_lab[j, :n[j]].view(np.uint32), r["labels"]), (rounds, j, "labels")
            assert np.array_equal(h_oi[j, :no].view(np.uint32), r["obstacle_idx"]), (rounds, j, "obstacle_idx")
            assert np.array_equal(h_cl[j, :no], r["cluster_labels"]), (rounds, j, "cluster labels")
        rounds += 1
    print(f"soak ok ({C} contexts x {B} frames{', overlapped tails' if overlap else ''}): {rounds} rounds x 3 x {F} frames, "
          "every output identical to the single-frame path")
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
