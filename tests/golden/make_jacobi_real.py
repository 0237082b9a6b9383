"""Regenerates tests/golden/jacobi_real.json: the restated Eigen 3.4 JacobiSVD<Matrix3f> (oracle/lidar_oracle.c,
SURVEY Appendix A; what the device reproduces bit for bit) checked on the covariances REAL data produces -- every plane
fit the oracle makes on all 154 frames x the two committed configurations, BASELINE configs[0] (P3 I3), configs[2]
(1M-point cloud) and configs[4] (5M-point cloud) -- against float64 LAPACK (numpy.linalg.eigh of the same float32
covariance): largest difference of the unit normal, sign convention (the normal points up: c > 0, every time),
smallest relative eigen-gap between the two smallest eigenvalues (what the accuracy of the normal hangs on) and the
largest number of sweeps the Jacobi loop took.  tests/test_stream.py recomputes part of it and asserts the bounds.

This widens the only lever the image leaves on the segmentation half (Eigen itself is absent): the
solve had so far only been compared with numpy on synthetic matrices."""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from util import STREAM_CONFIGS, load_frame, load_stream_frame, stream_names, synthetic_scene  # noqa: E402


def check_fits(records):
    """(max |normal - eigh normal|, min relative gap, max sweeps, fits, fits with c <= 0, failed fits, max of
    |difference| x gap / float32 epsilon) of the recorded fits"""
    worst, gap, sweeps, bad_sign, failed, cond = 0.0, np.inf, 0, 0, 0, 0.0
    for r in records:
        if r[15] != 0:
            failed += 1
            continue
        cov = r[:9].astype(np.float64).reshape(3, 3)
        w, v = np.linalg.eigh(cov)  # ascending eigenvalues
        n64 = v[:, 0]
        normal = r[9:12].astype(np.float64)
        if np.dot(n64, normal) < 0:
            n64 = -n64
        diff = float(np.abs(n64 - normal).max())
        g = float((w[1] - w[0]) / max(abs(w[2]), 1e-300))
        worst = max(worst, diff)
        gap = min(gap, g)
        cond = max(cond, diff * g / float(np.finfo(np.float32).eps))  # a float32 solve owes about eps / gap
        sweeps = max(sweeps, int(r[13]))
        bad_sign += int(not r[11] > 0)
    return worst, gap, sweeps, len(records), bad_sign, failed, cond


def frame_job(args):
    name, cname = args
    skw, _ = STREAM_CONFIGS[cname]
    with oracle.FitTrace() as tr:
        r = oracle.segment(load_stream_frame(name), oracle.SegCfg(**skw))
    assert r["rc"] == 0
    return cname, check_fits(tr.records)


def merge(rows):
    rows = list(rows)
    return {"fits": int(sum(r[3] for r in rows)), "max_abs_normal_diff_vs_float64_eigh": max(r[0] for r in rows),
            "min_relative_eigen_gap": min(r[1] for r in rows), "max_jacobi_sweeps": max(r[2] for r in rows),
            "fits_with_c_not_positive": int(sum(r[4] for r in rows)), "failed_fits": int(sum(r[5] for r in rows)),
            "max_diff_times_gap_over_eps32": max(r[6] for r in rows)}


def single(pts, skw):
    with oracle.FitTrace() as tr:
        r = oracle.segment(pts, oracle.SegCfg(**skw))
    assert r["rc"] == 0
    return merge([check_fits(tr.records)])


def main():
    out = {}
    jobs = [(n, c) for c in STREAM_CONFIGS for n in stream_names()]
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(frame_job, jobs, chunksize=4)
    for cname in STREAM_CONFIGS:
        out[f"stream_{cname}"] = merge(r for c, r in res if c == cname)
    out["configs0_p3i3"] = single(load_frame("0000000000"), dict(number_of_planar_partitions=3, number_of_iterations=3))
    out["configs2_synth1m"] = single(synthetic_scene(600_000, 2000, 200, 20240601),
                                     dict(number_of_planar_partitions=12, number_of_iterations=3))
    out["configs4_synth5m"] = single(synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0),
                                     dict(number_of_planar_partitions=24, number_of_iterations=3))
    # what tests/test_stream.py asserts: the solve is as accurate as float32 allows on every real covariance (its error
    # times the eigen-gap stays within a few float32 epsilons), the normal always points up, no fit fails, at most 4
    # sweeps; in absolute terms within 1e-4 of the float64 eigenvector except on configs[4], whose 8 m x 200 m ground
    # strips leave the tilt about the long axis conditioned at 1.2e-3 (1.7e-4 there)
    out["bounds_asserted"] = {"max_diff_times_gap_over_eps32": 4.0, "fits_with_c_not_positive": 0, "failed_fits": 0,
                              "max_jacobi_sweeps": 4,
                              "max_abs_normal_diff_vs_float64_eigh": {"default": 1e-4, "configs4_synth5m": 2.5e-4}}
    for k, v in out.items():
        print(k, v)
    with open(os.path.join(ROOT, "tests", "golden", "jacobi_real.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
