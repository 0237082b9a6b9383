"""Per-stage HIP-event timing of the hot path on one frame (diagnostic; not the bench)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from util import load_frame, synthetic_scene  # noqa: E402


def run(name, pts, scfg, ccfg, reps=5):
    ctx = Context(0)
    ctx.reserve(pts.shape[0])
    out = ctx.segment_cluster(pts, scfg, ccfg)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ctx.segment_cluster(pts, scfg, ccfg)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.segment_cluster(pts, scfg, ccfg)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    tot = sum(v[0] for v in prof.values()) / reps
    print(f"== {name}: n={pts.shape[0]} ground={len(out['ground_idx'])} obstacle={len(out['obstacle_idx'])} "
          f"clusters={out['n_clusters']}  host API {host_ms:.3f} ms/frame, sum of stages {tot:.3f} ms")
    for k, (ms, cnt) in prof.items():
        print(f"   {k:14s} {ms / reps:9.4f} ms  ({cnt // reps} event pairs)")
    ctx.close()


if __name__ == "__main__":
    import torch
    print(torch.cuda.get_device_name(0), torch.version.hip)
    which = sys.argv[1:] or ["c2"]
    if "c2" in which:
        run("C2 frame0 P6 I5 d2=0.25", load_frame("0000000000"),
            SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5),
            ClusteringConfiguration(0.25, 0.5))
        run("C2 frame153", load_frame("0000000153"),
            SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5),
            ClusteringConfiguration(0.25, 0.5))
    if "c3" in which:
        run("C3 synthetic 1M P12 I3 d2=0.09", synthetic_scene(600_000, 2000, 200, 20240601),
            SegmentationConfiguration(number_of_planar_partitions=12, number_of_iterations=3),
            ClusteringConfiguration(0.09, 0.5), reps=3)
    if "c5" in which:
        run("C5 synthetic 5M P24 I3 d2=0.04", synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0),
            SegmentationConfiguration(number_of_planar_partitions=24, number_of_iterations=3),
            ClusteringConfiguration(0.04, 0.5), reps=2)
