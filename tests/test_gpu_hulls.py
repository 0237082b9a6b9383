"""N3: per-cluster convex hulls on the device (lpx_cluster_hulls) against the C restatement
oracle.cluster_hulls / oracle.convex_hull (Andrew monotone chain, CCW; PARITY UNPINNED: the reference's
implementation lives in the absent Convex-Hull submodule)."""
import numpy as np
import pytest

import oracle
from lidar_processing_amd import ClusteringConfiguration, Clusterer, SegmentationConfiguration
from util import FRAMES, load_frame

pytestmark = pytest.mark.gpu


def check_hulls(ctx, obs, d2, q, mn=4, max_points=20):
    lab, nc = ctx.cluster(obs, ClusteringConfiguration(d2, q, mn))
    off, idx, xy = ctx.cluster_hulls(obs.shape[0], nc, max_points)
    woff, widx = oracle.cluster_hulls(obs, lab, nc, max_points)
    assert np.array_equal(off, woff), (off[:10], woff[:10])
    assert np.array_equal(idx, widx)
    assert np.array_equal(xy.view(np.uint32), (obs[widx, :2] + np.float32(0)).view(np.uint32))
    return lab, nc, off, idx, xy


@pytest.mark.parametrize("frame", FRAMES)
@pytest.mark.parametrize("max_points", [20, 64, 2 ** 32 - 1])
def test_hulls_of_real_frames(ctx, frame, max_points):
    """max_points = 20: the reference's convex branch; 2^32 - 1: a hull for every cluster (findOrderedConvexOutlines)"""
    pts = load_frame(frame)
    obs = pts[oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6, number_of_iterations=5))["obstacle_idx"]]
    lab, nc, off, idx, xy = check_hulls(ctx, obs, 0.25, 0.5, max_points=max_points)
    sizes = np.bincount(lab[lab >= 0], minlength=nc)
    small = sizes < max_points
    assert ((np.diff(off) > 0) == small).all()  # exactly the clusters below the limit have a hull
    # counter-clockwise, convex, and every member inside: checked in float64 on a sample of clusters
    for c in np.nonzero(small)[0][:60]:
        h = xy[off[c]:off[c + 1]].astype(np.float64)
        if len(h) >= 3:
            x, y = h[:, 0], h[:, 1]
            assert 0.5 * np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y) > 0
            mem = obs[lab == c, :2].astype(np.float64)
            for a, b in zip(h, np.roll(h, -1, axis=0)):
                cr = (b[0] - a[0]) * (mem[:, 1] - a[1]) - (b[1] - a[1]) * (mem[:, 0] - a[0])
                assert (cr >= -1e-6).all()


def test_hull_known_answers(ctx):
    """square with interior and edge points, collinear points, duplicates, 1 to 3 points, -0.0"""
    groups = [
        [(0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5), (0.5, 0), (1, 0.5), (0, 0), (1, 1)],   # -> the 4 corners
        [(0, 0), (1, 1), (2, 2), (3, 3), (1.5, 1.5)],                                        # collinear -> 2 ends
        [(5, 5)] * 6,                                                                        # identical -> 1 point
        [(0, 0), (0, 0), (2, 0), (2, 0)],                                                    # 2 distinct points
        [(-0.0, 0.0), (0.0, -0.0), (1, 0), (0, 1)],                                          # signed zeros: triangle
        [(0, 0), (4, 0), (2, 3)],                                                            # triangle
        [(0, 0), (4, 0), (2, -3)],                                                           # triangle, other turn
        [(0, 0), (1, 2), (2, 0), (1, -2), (1, 0), (1, 1), (1, -1)],                          # rhombus + axis points
    ]
    want = [[(0, 0), (1, 0), (1, 1), (0, 1)], [(0, 0), (3, 3)], [(5, 5)], [(0, 0), (2, 0)],
            [(0, 0), (1, 0), (0, 1)], [(0, 0), (4, 0), (2, 3)], [(0, 0), (2, -3), (4, 0)],
            [(0, 0), (1, -2), (2, 0), (1, 2)]]
    obs, lab_true = [], []
    for g, pts in enumerate(groups):
        for (x, y) in pts:
            obs.append((x + 100.0 * g, y, 0.0, 0.0))
            lab_true.append(g)
    obs = np.array(obs, np.float32)
    clu = Clusterer(context=ctx)
    clu.update_configuration(ClusteringConfiguration(25.0, 1.0, 1))  # d = 5 m: one cluster per group
    lab = clu.cluster(obs)
    assert np.array_equal(lab, np.array(lab_true))
    outlines = clu.convex_outlines(obs)
    assert len(outlines) == len(groups)
    for g, (got, w) in enumerate(zip(outlines, want)):
        assert [(float(x) - 100.0 * g, float(y)) for x, y in got] == [(float(a), float(b)) for a, b in w], g
    lab, nc, off, idx, xy = check_hulls(ctx, obs, 25.0, 1.0, mn=1)
    for g in range(len(groups)):  # the kept duplicate is the one with the smallest index
        first = {}
        for i in np.nonzero(np.array(lab_true) == g)[0]:
            first.setdefault((float(obs[i, 0] + 0.0), float(obs[i, 1] + 0.0)), int(i))
        assert set(idx[off[g]:off[g + 1]].tolist()) <= set(first.values())


@pytest.mark.parametrize("seed", range(4))
def test_hulls_random_against_scipy(ctx, seed):
    """independent check of the vertex SETS against scipy (qhull) on float32 Gaussian blobs"""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    k, per = 40, rng.integers(3, 19, 40)
    obs = np.zeros((int(per.sum()), 4), np.float32)
    lab_true = np.repeat(np.arange(k), per)
    obs[:, 0] = 50.0 * lab_true + rng.normal(0, 0.6, obs.shape[0])
    obs[:, 1] = rng.normal(0, 0.6, obs.shape[0])
    lab, nc, off, idx, xy = check_hulls(ctx, obs, 36.0, 1.0, mn=1)
    assert nc == k and np.array_equal(lab, lab_true)
    for c in range(k):
        mem = np.nonzero(lab_true == c)[0]
        sv = set(mem[ConvexHull(obs[mem, :2].astype(np.float64)).vertices].tolist())
        assert set(idx[off[c]:off[c + 1]].tolist()) == sv


def test_hulls_edge_cases(ctx):
    one = np.zeros((3, 4), np.float32)
    one[:, 0] = [0, 10, 20]
    lab, nc = ctx.cluster(one, ClusteringConfiguration(0.18, 0.5, 4))  # everything rejected: no cluster
    off, idx, xy = ctx.cluster_hulls(3, nc)
    assert nc == 0 and off.tolist() == [0] and idx.size == 0
    lab, nc = ctx.cluster(one, ClusteringConfiguration(0.18, 0.5, 1))  # three singletons
    off, idx, xy = ctx.cluster_hulls(3, nc)
    assert off.tolist() == [0, 1, 2, 3] and idx.tolist() == [0, 1, 2]
    off, idx, xy = ctx.cluster_hulls(3, nc, max_points=1)  # nothing is below the limit
    assert off.tolist() == [0, 0, 0, 0]
    # fused path: hulls of the clusters of the obstacle cloud kept on the device
    pts = load_frame(FRAMES[2])[:60_000]
    out = ctx.segment_cluster(pts, SegmentationConfiguration(), ClusteringConfiguration())
    obs = pts[out["obstacle_idx"]]
    off, idx, xy = ctx.cluster_hulls(obs.shape[0], out["n_clusters"])
    woff, widx = oracle.cluster_hulls(obs, out["cluster_labels"], out["n_clusters"])
    assert np.array_equal(off, woff) and np.array_equal(idx, widx)
