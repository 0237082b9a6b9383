"""Helper of tests/test_gpu_pipeline.py::test_point_states_in_hbm_give_the_same_labels: run with LPX_RP_STATE=2 | 3
and LPX_RS_STATE=1 in the environment (read once per process) -- the replay kernels then keep their point states per
component in LDS by member position (2) or in HBM, one byte per point (3; the search replay: LPX_RS_STATE), whatever the
frame size: the paths that otherwise only frames beyond the whole-cloud LDS bitmap take.  Real frames in
both neighbour modes against the C restatement, and a ragged batch against the single-frame path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from test_gpu_batch import check_frame, run_batch, single  # noqa: E402
from test_gpu_pipeline import check_against_oracle  # noqa: E402
from util import FRAMES, load_frame, synthetic_scene  # noqa: E402

assert os.environ.get("LPX_RP_STATE") in ("1", "2", "3") and os.environ.get("LPX_RS_STATE") == "1"
from lidar_processing_amd import _lib as _l  # noqa: E402
assert b"development build" in _l.lib().lpx_build_info(), "the knobs are read by liblpx_dev.so only (LPX_LIB)"
skw = dict(number_of_planar_partitions=6, number_of_iterations=5)
checked = 0
for mode in ("lists", "search"):
    ctx = Context(0)
    ctx.set_neighbour_mode(mode)
    try:
        for f in FRAMES:
            pts = load_frame(f)
            for d2, q in ((0.25, 0.5), (0.18, 1.0), (0.25, 0.0)):
                out = ctx.segment_cluster(pts, SegmentationConfiguration(**skw), ClusteringConfiguration(d2, q))
                check_against_oracle(out, pts, oracle.SegCfg(**skw), oracle.CluCfg(d2, q))
                checked += 1
    finally:
        ctx.close()
# one component of more members than the component-local bitmap holds (65 536): 120 000 points scattered over a
# 90 m x 90 m sheet (12.6 per m2, far above the percolation density for d = 0.5 m: the largest component has 119 993
# members) -- LPX_RP_STATE=2 sends it through the HBM form INSIDE the local-state kernel
rng = np.random.default_rng(11)
lattice = np.zeros((120_000, 4), np.float32)
lattice[:, 0] = rng.uniform(0, 90, lattice.shape[0]).astype(np.float32)
lattice[:, 1] = rng.uniform(0, 90, lattice.shape[0]).astype(np.float32)
lattice[:, 2] = rng.uniform(-0.05, 0.05, lattice.shape[0]).astype(np.float32)
want_l, want_n = oracle.cluster(lattice, oracle.CluCfg(0.25, 0.5))
ctx = Context(0)
ctx.set_neighbour_mode("lists")
try:
    lab, nc = ctx.cluster(lattice, ClusteringConfiguration(0.25, 0.5))
    assert nc == want_n and np.array_equal(lab, want_l), "sheet: one component of 119 993 members"
    checked += 1
finally:
    ctx.close()
seg_kw = dict(number_of_planar_partitions=4, number_of_iterations=3)
clu_kw = dict(distance_squared=0.36, cluster_quality=0.3, min_cluster_size=3)
sizes = [30_000, 0, 3, 11_111, 8_193, 64, 20_001, 5_000, 257]
clouds = []
for i, n in enumerate(sizes):
    base = synthetic_scene(max(n, 64) - max(n, 64) // 3, 8, max(1, (max(n, 64) // 3) // 8), seed=500 + i)
    clouds.append(base[:n])
one = Context(0)
refs = [single(one, c, seg_kw, clu_kw) for c in clouds]
one.close()
for mode in ("search", "lists"):
    bctx = Context(0, batch=len(clouds))
    bctx.set_neighbour_mode(mode)
    try:
        for res, ref in zip(run_batch(bctx, clouds, seg_kw, clu_kw), refs):
            check_frame(res, ref)
            checked += 1
    finally:
        bctx.close()
print("state check ok:", checked, "cases")
