"""Round 6: which path gives a different partition now and then?  The 200k-point cloud of test_ragged_batch_with_long_segments
through every path, REPS times, against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration  # noqa: E402
from test_gpu_batch import run_batch  # noqa: E402
from util import FRAMES, load_frame  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
big = np.concatenate([load_frame(f) for f in FRAMES])[:200_000]
clouds = [big, load_frame(FRAMES[0])[:1], load_frame(FRAMES[1])[:100], load_frame(FRAMES[2])[:30_000], big[:60_001]]
seg = dict(number_of_planar_partitions=2, number_of_iterations=3)
clu = dict(distance_squared=0.25, cluster_quality=0.5)
want = []
for c in clouds:
    r = oracle.segment(c, oracle.SegCfg(**seg))
    lab, n = oracle.cluster(c[r["obstacle_idx"]], oracle.CluCfg(0.25, 0.5))
    want.append((r, lab, n))
bad = {}
POISON = os.environ.get("R6_POISON", "1") == "1"
for mode in ("search", "lists"):
    one = Context(0)
    one.set_neighbour_mode(mode)
    if POISON:
        # what a long test session leaves on a shared context: a 5M-point frame (capacity, statistics), the stage hooks
        from util import synthetic_scene
        pts5 = synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0)
        one.segment_cluster(pts5, SegmentationConfiguration(number_of_planar_partitions=24, number_of_iterations=3),
                            ClusteringConfiguration(0.04, 0.5))
        xyz = np.ascontiguousarray(big[:50_000, :3])
        one.dbg_components(xyz, 0.25)
        one.dbg_neighbours(xyz[:5000], 0.25)
        one.dbg_kd_layout(xyz)
        one.dbg_plane(xyz)
    for rep in range(REPS):
        for ci in (0, 4, 3):
            out = one.segment_cluster(clouds[ci], SegmentationConfiguration(**seg), ClusteringConfiguration(**clu))
            ok = np.array_equal(out["cluster_labels"], want[ci][1]) and np.array_equal(out["labels"], want[ci][0]["labels"])
            if not ok:
                bad[("single", mode, ci)] = bad.get(("single", mode, ci), 0) + 1
    one.close()
    b = Context(0, batch=len(clouds))
    b.set_neighbour_mode(mode)
    for rep in range(REPS):
        res = run_batch(b, clouds, seg, clu)
        for ci, r in enumerate(res):
            ok = r["status"] == 0 and np.array_equal(r["cluster_labels"], want[ci][1]) and \
                np.array_equal(r["labels"], want[ci][0]["labels"])
            if not ok:
                bad[("batch", mode, ci)] = bad.get(("batch", mode, ci), 0) + 1
    b.close()
print("reps", REPS, "env", {k: v for k, v in os.environ.items() if k.startswith("LPX_")}, "mismatches", bad)
# fresh batch contexts, lists mode: the first call runs on the default workspace (single-pass region partly exhausted:
# some groups reserve, others count -- which ones depends on scheduling; the result must not)
fresh_bad = 0
for rep in range(int(os.environ.get("R6_FRESH", "60"))):
    b = Context(0, batch=len(clouds))
    b.set_neighbour_mode("lists")
    if os.environ.get("R6_WS"):  # "nb,rs": words per point of the two list regions, set explicitly (any library)
        nb_w, rs_w = (int(v) for v in os.environ["R6_WS"].split(","))
        b.reserve(max(c.shape[0] for c in clouds) + 7, nb_w)
        b.reserve_single_pass(rs_w)
    try:
        res = run_batch(b, clouds, seg, clu)
        for ci, r in enumerate(res):
            if r["status"] != 0 or not np.array_equal(r["cluster_labels"], want[ci][1]):
                fresh_bad += 1
                d = np.nonzero(r["cluster_labels"] != want[ci][1])[0] if r["status"] == 0 else []
                print("fresh batch rep", rep, "cloud", ci, "status", r["status"], "points differing", len(d), d[:6])
    finally:
        b.close()
print("fresh lists batch contexts: mismatches", fresh_bad)
