"""Regenerates tests/golden/synth_golden.json: the goldens of BASELINE.json configs[2] (1M-point cloud) and configs[4]
(5M-point cloud) at their full sizes -- counts and CRC-32 of the segmentation labels, the obstacle order, the cluster
labels and the plane words, the same seven columns as stream_golden.npz.  The clouds come from the seeded generator of
tests/util.py, so only the rows travel.  Cluster labels are those of the REFERENCE's own kdtree.hpp +
queue.hpp build (oracle.ref_fec, oracle/_ref/libkdref.so) and the restatement is asserted equal to them; the
segmentation half is the oracle's canonical output (Eigen absent: regression lock).

bench.py compares every frame of its timed region's last step with these rows (`verified` of its line): it may not call
the oracle itself.  Build container only (minutes of CPU)."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from util import synthetic_scene  # noqa: E402

CASES = {
    "synth1m": (dict(n_ground=600_000, n_boxes=2000, pts_per_box=200, seed=20240601),
                dict(number_of_planar_partitions=12, number_of_iterations=3), dict(distance_squared=0.09, cluster_quality=0.5)),
    "synth5m": (dict(n_ground=2_000_000, n_boxes=3000, pts_per_box=1000, seed=20240602, extent=100.0),
                dict(number_of_planar_partitions=24, number_of_iterations=3), dict(distance_squared=0.04, cluster_quality=0.5)),
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    out = {}
    for name, (gen, skw, ckw) in CASES.items():
        pts = synthetic_scene(**gen)
        r = oracle.segment(pts, oracle.SegCfg(**skw))
        assert r["rc"] == 0
        obs = pts[r["obstacle_idx"]]
        ccfg = oracle.CluCfg(**ckw)
        lab_ref, nc_ref = oracle.ref_fec(obs, ccfg)
        lab, nc = oracle.cluster(obs, ccfg)
        assert nc == nc_ref and np.array_equal(lab, lab_ref), f"{name}: oracle != reference build"
        out[name] = {"generator": gen, "seg": skw, "clu": ckw, "n": int(pts.shape[0]),
                     "row": [len(r["ground_idx"]), len(r["obstacle_idx"]), int(nc_ref),
                             crc(r["labels"].astype(np.uint8)), crc(r["obstacle_idx"]), crc(lab_ref), crc(r["planes"])],
                     "columns": ["n_ground", "n_obstacle", "n_clusters", "crc32(labels as u8)", "crc32(obstacle_idx u32)",
                                 "crc32(cluster_labels i32, reference build)", "crc32(planes f32)"]}
        print(name, out[name]["row"], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "synth_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
