"""Regenerates tests/golden/sort_bound.json: what the UNSPECIFIED tie order of the reference's x sort
(src/segmentation.cpp:114-122, std::sort(std::execution::par, ...) on indices compared by x alone: SURVEY H2) can change.

The repository's canonical order is (x, index) -- a stable sort.  A build of the reference orders points of EQUAL x
differently: without TBB headers the call is libstdc++'s serial std::sort (introsort), with the TBB backend leaves of at
most 500 indices are sorted by std::sort and merged stably (oracle/sort_order.cpp restates both on this image's
libstdc++ 11.4).  Only ties that straddle a segment boundary matter: those points land in the neighbouring segment, are
tested against ITS planes, and shift every later point of the output clouds.  Per configuration: frames with at least
one point in another segment, the most such points in a frame, and the most segmentation labels that differ from the
canonical result in a frame (the oracle run on the cloud re-indexed in the other order, mapped back).

The KITTI frames are quantised to 1 mm, so ties in x are common (hundreds per frame) but a boundary falls inside a tie
group of only a few points."""
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

MODES = ("serial", "tbb")


def compare(pts, skw):
    """{mode: (points in another segment, labels that differ, planes max |delta|)} against the canonical order"""
    cfg = oracle.SegCfg(**skw)
    P = skw["number_of_planar_partitions"]
    n = pts.shape[0]
    n_per = n // P
    canon = oracle.x_sort_order(pts[:, 0], "stable")
    seg_c = np.full(n, -1, np.int64)
    seg_c[canon[:P * n_per]] = np.arange(P * n_per) // max(n_per, 1)
    base = oracle.segment(pts, cfg)
    out = {}
    for mode in MODES:
        order = oracle.x_sort_order(pts[:, 0], mode)
        assert np.array_equal(pts[order, 0], pts[canon, 0])  # the same sorted x: only ties move
        seg_m = np.full(n, -1, np.int64)
        seg_m[order[:P * n_per]] = np.arange(P * n_per) // max(n_per, 1)
        moved = int((seg_m != seg_c).sum())
        # the cloud re-indexed in that order: its canonical (x, index) sort IS that order
        r = oracle.segment(np.ascontiguousarray(pts[order]), cfg)
        labels = np.zeros(n, np.uint32)
        labels[order] = r["labels"]
        out[mode] = (moved, int((labels != base["labels"]).sum()),
                     float(np.abs(r["planes"].astype(np.float64) - base["planes"].astype(np.float64)).max()))
    return out


def frame_job(args):
    name, cname = args
    return cname, compare(load_stream_frame(name), STREAM_CONFIGS[cname][0])


def merge(rows):
    rows = list(rows)
    return {mode: {"frames": len(rows), "frames_with_a_point_in_another_segment": int(sum(r[mode][0] > 0 for r in rows)),
                   "points_in_another_segment_per_frame_max": int(max(r[mode][0] for r in rows)),
                   "labels_that_differ_per_frame_max": int(max(r[mode][1] for r in rows)),
                   "labels_that_differ_total": int(sum(r[mode][1] for r in rows)),
                   "plane_coefficient_delta_max": float(max(r[mode][2] for r in rows))} for mode in MODES}


def compute(frames=None):
    names = stream_names() if frames is None else frames
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(frame_job, [(n, c) for c in STREAM_CONFIGS for n in names], chunksize=4)
    out = {f"stream_{c}": merge(r for cc, r in res if cc == c) for c in STREAM_CONFIGS}
    if frames is None:
        out["configs0_p3i3"] = merge([compare(load_frame("0000000000"), dict(number_of_planar_partitions=3, number_of_iterations=3))])
        out["configs2_synth1m"] = merge([compare(synthetic_scene(600_000, 2000, 200, 20240601),
                                                 dict(number_of_planar_partitions=12, number_of_iterations=3))])
        out["configs4_synth5m"] = merge([compare(synthetic_scene(2_000_000, 3000, 1000, 20240602, extent=100.0),
                                                 dict(number_of_planar_partitions=24, number_of_iterations=3))])
    return out


def main():
    out = compute()
    for k, v in out.items():
        print(k, json.dumps(v))
    with open(os.path.join(ROOT, "tests", "golden", "sort_bound.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
