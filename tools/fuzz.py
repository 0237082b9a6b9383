"""Randomised parity fuzz: random scenes (synthetic plane + boxes, KITTI frame slices, random sub-samples) and random
configurations through the single-frame host path (lists and search), the device path and multi-frame chains on
long-lived contexts, every output compared with the oracle.  usage: fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame, synthetic_scene
from test_gpu_batch import run_batch

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kitti = [load_frame(f) for f in FRAMES]


def scene():
    kind = rng.integers(0, 7)
    if kind == 4:  # tiny clouds, fewer points than partitions included
        n = int(rng.integers(0, 200))
        c = np.zeros((n, 4), np.float32)
        c[:, :3] = rng.normal(0.0, 2.0, (n, 3)).astype(np.float32)
        c[:, 2] = c[:, 2] * 0.3 - 1.5
        return c
    if kind == 5:  # a frame slice with returns far outside the fixed-point range of the moments
        f = kitti[rng.integers(0, len(kitti))]
        c = f[: int(rng.integers(5_000, 60_000))].copy()
        for _ in range(int(rng.integers(1, 6))):
            c[rng.integers(0, c.shape[0]), rng.integers(0, 3)] = float(rng.choice([-1, 1])) * float(rng.uniform(2.1e3, 9.0e5))
        return c
    if kind == 6:  # heavy ties and exact duplicates: coordinates on a 0.1 m lattice
        f = kitti[rng.integers(0, len(kitti))]
        c = f[rng.random(f.shape[0]) < 0.3].copy()
        c[:, :3] = np.round(c[:, :3] * 10.0) / 10.0
        return c
    if kind == 0:
        n = int(rng.integers(2_000, 150_000))
        nb = int(rng.integers(1, 120))
        return synthetic_scene(n, nb, int(rng.integers(20, 900)), seed=int(rng.integers(1, 1 << 30)),
                               extent=float(rng.choice([20.0, 60.0, 120.0])))
    f = kitti[rng.integers(0, len(kitti))]
    if kind == 1:
        a = int(rng.integers(0, f.shape[0] // 2))
        return f[a:a + int(rng.integers(1_000, f.shape[0] - a))].copy()
    if kind == 2:
        keep = rng.random(f.shape[0]) < rng.uniform(0.05, 1.0)
        return f[keep].copy()
    return f.copy()


def expect(c, seg_kw, clu_kw):
    o = oracle.segment(c, oracle.SegCfg(**seg_kw))
    lab, nc = oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**clu_kw))
    return o, lab, nc


def same(tag, r, o, lab, nc, info):
    ok = (np.array_equal(r["labels"], o["labels"]) and np.array_equal(r["ground_idx"], o["ground_idx"]) and
          np.array_equal(r["obstacle_idx"], o["obstacle_idx"]) and
          np.array_equal(np.asarray(r["planes"]).view(np.uint32).ravel(), o["planes"].view(np.uint32).ravel()) and
          np.array_equal(r["cluster_labels"], lab) and r["n_clusters"] == nc)
    if not ok:
        d = np.flatnonzero(r["cluster_labels"] != lab) if r["cluster_labels"].shape == lab.shape else []
        print("MISMATCH", tag, info, "cluster label diffs", len(d), "n_clusters", r["n_clusters"], nc, flush=True)
    return ok


ctx_l, ctx_s = Context(0), Context(0)
ctx_s.set_neighbour_mode("search")
bctx = {m: Context(0, batch=3) for m in ("lists", "search")}
for m, c in bctx.items():
    c.set_neighbour_mode(m)
t_end = time.time() + seconds
cases = bad = 0
while time.time() < t_end:
    seg_kw = dict(number_of_planar_partitions=int(rng.integers(1, 9)), number_of_iterations=int(rng.integers(1, 6)))
    clu_kw = dict(distance_squared=float(rng.choice([0.04, 0.09, 0.18, 0.25, 0.49, 1.0])),
                  cluster_quality=float(rng.choice([0.0, 0.3, 0.5, 0.8, 1.0])),
                  min_cluster_size=int(rng.choice([1, 4, 10])))
    clouds = [scene() for _ in range(3)]
    want = [expect(c, seg_kw, clu_kw) for c in clouds]
    scfg, ccfg = SegmentationConfiguration(**seg_kw), ClusteringConfiguration(**clu_kw)
    info = (seg_kw, clu_kw, [c.shape[0] for c in clouds])
    for c, (o, lab, nc) in zip(clouds, want):
        bad += not same("single lists", ctx_l.segment_cluster(c, scfg, ccfg), o, lab, nc, info)
        bad += not same("single search", ctx_s.segment_cluster(c, scfg, ccfg), o, lab, nc, info)
    for m, bc in bctx.items():
        for r, (o, lab, nc) in zip(run_batch(bc, clouds, seg_kw, clu_kw), want):
            if r["status"] == 3:  # LPX_ERR_CAPACITY of the list path in a chain: the caller's retry, not a mismatch
                continue
            bad += not same("batch " + m, r, o, lab, nc, info)
    cases += 1
print("fuzz cases", cases, "mismatches", bad, flush=True)
sys.exit(1 if bad else 0)
