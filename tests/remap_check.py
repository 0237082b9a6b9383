"""Helper of tests/test_gpu_batch.py::test_xcd_affine_launch_geometry_is_a_bijection: run with LPX_REMAP=<mask> in the
environment (the library reads it once per process).  Ragged batches of 8, 13, 16 and 19 frames -- whole groups of eight
re-read, the rest identity -- in both neighbour modes and, for the lists mode kernels, on a batch context; every frame
against the single-frame path (gridDim.z = 1: never re-read)."""
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from test_gpu_batch import check_frame, run_batch, single  # noqa: E402
from util import synthetic_scene  # noqa: E402

seg_kw = dict(number_of_planar_partitions=4, number_of_iterations=3)
clu_kw = dict(distance_squared=0.36, cluster_quality=0.3, min_cluster_size=3)
sizes = [30_000, 0, 3, 11_111, 4096 * 2 + 1, 64, 20_001, 5_000, 17_000, 257, 9_000, 33_333, 1_000, 12_345, 8_191, 8_193,
         2_048, 25_000, 777]
clouds = []
for i, n in enumerate(sizes):
    base = synthetic_scene(max(n, 64) - max(n, 64) // 3, 8, max(1, (max(n, 64) // 3) // 8), seed=300 + i)
    clouds.append(base[:n])
one = Context(0)
refs = [single(one, c, seg_kw, clu_kw) for c in clouds]
one.close()
checked = 0
for mode in ("search", "lists"):
    for B in (8, 13, 16, 19):
        bctx = Context(0, batch=B)
        bctx.set_neighbour_mode(mode)
        try:
            for res, ref in zip(run_batch(bctx, clouds[:B], seg_kw, clu_kw), refs[:B]):
                check_frame(res, ref)
                checked += 1
            for res, ref in zip(run_batch(bctx, clouds[:B][::-1], seg_kw, clu_kw)[::-1], refs[:B]):
                check_frame(res, ref)
                checked += 1
        finally:
            bctx.close()
print("remap check ok:", os.environ.get("LPX_REMAP"), checked, "frames")
