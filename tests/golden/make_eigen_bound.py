"""Generates tests/golden/eigen_bound.json: how far the oracle's canonical plane fit (exact integer moments) sits
from a float32 evaluation of src/segmentation.cpp:62-102,:287-307 in every summation order of tests/eigen_like.py
(sequential, numpy pairwise, 4- and 8-lane packet accumulation with a horizontal add, blocked-GEMM depth blocks,
and both association orders of the per-row dot), over the five BASELINE configurations:

  stream_p6i5 / stream_p2i3   all 154 data/*.pcd frames (committed under tests/golden/stream/)
  c1_p3i3                     configs[0]: the three committed KITTI frames, 3 segments, 3 iterations
  synth1m_p12i3               configs[2]: the 1M-point plane + boxes cloud, 12 segments
  synth5m_p24i3               configs[4]: the 5M-point cloud, 24 segments

Per (configuration, order): the largest normal-component delta, the largest |d| delta, the most points of one frame
whose ground / obstacle side differs, the total of such points and the number of frames.  Segmentation parity stays
UNPINNED (Eigen is not in the image); this table is the committed bound of that gap, asserted by
tests/test_stream.py::test_canonical_moments_stay_within_tolerance_of_float32_eigen_order.

    python tests/golden/make_eigen_bound.py        (about two minutes on 8 cores)
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from eigen_like import ORDERS, segment_float32  # noqa: E402
from util import FRAMES, STREAM_CONFIGS, load_frame, load_stream_frame, stream_names, synthetic_scene  # noqa: E402

CONFIGS = {
    "stream_p6i5": ("stream", STREAM_CONFIGS["p6i5_d025q05"][0]),
    "stream_p2i3": ("stream", STREAM_CONFIGS["p2i3_d018q05"][0]),
    "c1_p3i3": ("kitti3", dict(number_of_planar_partitions=3, number_of_iterations=3)),
    "synth1m_p12i3": ("synth1m", dict(number_of_planar_partitions=12, number_of_iterations=3)),
    "synth5m_p24i3": ("synth5m", dict(number_of_planar_partitions=24, number_of_iterations=3)),
}


def frames_of(kind):
    if kind == "stream":
        return [("stream", n) for n in stream_names()]
    if kind == "kitti3":
        return [("kitti3", n) for n in FRAMES]
    return [(kind, kind)]


def load(kind, name):
    if kind == "stream":
        return load_stream_frame(name)
    if kind == "kitti3":
        return load_frame(name)
    if kind == "synth1m":
        return synthetic_scene(600_000, 2000, 200, 20240601)
    return synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0)


def one_frame(job):
    """rows (config, order, normal delta, d delta, flipped points) of one frame for every configuration on it"""
    kind, name, cnames = job
    pts = load(kind, name)
    rows = []
    for cname in cnames:
        cfg = oracle.SegCfg(**CONFIGS[cname][1])
        r = oracle.segment(pts, cfg)
        for order in ORDERS:
            lab, pl = segment_float32(pts, cfg, order)
            rows.append((cname, order, float(np.abs(pl[:, :3] - r["planes"][:, :3]).max()),
                         float(np.abs(pl[:, 3] - r["planes"][:, 3]).max()), int((lab != r["labels"]).sum())))
    return rows


def jobs(stream_step=1):
    out = []
    for kind in ("stream", "kitti3", "synth1m", "synth5m"):
        cn = [c for c, (k, _) in CONFIGS.items() if k == kind]
        fr = frames_of(kind)
        if kind == "stream":
            fr = fr[::stream_step]
        out += [(k, n, cn) for k, n in fr]
    return out


def table(rows):
    t = {}
    for cname, order, dn, dd, fl in rows:
        e = t.setdefault(cname, {}).setdefault(order, {"normal_delta_max": 0.0, "d_delta_max_m": 0.0,
                                                       "flipped_points_per_frame_max": 0, "flipped_points_total": 0,
                                                       "frames": 0})
        e["normal_delta_max"] = max(e["normal_delta_max"], dn)
        e["d_delta_max_m"] = max(e["d_delta_max_m"], dd)
        e["flipped_points_per_frame_max"] = max(e["flipped_points_per_frame_max"], fl)
        e["flipped_points_total"] += fl
        e["frames"] += 1
    return t


def compute(stream_step=1, workers=None):
    js = jobs(stream_step)
    js.sort(key=lambda j: 0 if j[0].startswith("synth") else 1)  # the long jobs first
    with Pool(workers or min(8, os.cpu_count() or 1)) as pool:
        rows = [r for res in pool.map(one_frame, js, chunksize=1) for r in res]
    return table(rows)


if __name__ == "__main__":
    t = compute()
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "orders": list(ORDERS),
           "tolerances": {"normal": 1e-4, "d_m": 1e-3}, "configs": t}
    with open(os.path.join(HERE, "eigen_bound.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for c, o in t.items():
        for k, e in o.items():
            print(f"{c:16s} {k:11s} normal {e['normal_delta_max']:.2e}  d {e['d_delta_max_m']:.2e}  "
                  f"flips/frame <= {e['flipped_points_per_frame_max']}  total {e['flipped_points_total']} / {e['frames']} frames")
