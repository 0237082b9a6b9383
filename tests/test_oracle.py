"""CPU tests of the parity oracle (oracle/): pinned against the golden vectors produced from the
reference's own kdtree.hpp / queue.hpp build, against that build itself where it is available
(oracle/_ref/libkdref.so travels as a prebuilt binary), and by known-answer tests."""
import numpy as np
import pytest

import oracle
from util import FRAMES, brute_components, gold, load_frame, partition_signature, synthetic_scene

SEG_CFGS = {"p2i3": dict(number_of_planar_partitions=2, number_of_iterations=3),
            "p3i3": dict(number_of_planar_partitions=3, number_of_iterations=3),
            "p6i5": dict(number_of_planar_partitions=6, number_of_iterations=5)}
CLU_CFGS = {"d018q05": (0.18, 0.5), "d025q05": (0.25, 0.5), "d018q10": (0.18, 1.0)}

needs_ref = pytest.mark.skipif(oracle.ref() is None, reason="oracle/_ref/libkdref.so not built")


def test_frames_fixture_matches_reference_counts():
    # SURVEY 2 row 16 / BASELINE.md: 123 398 points in frame 0, 98 533 in frame 153
    assert load_frame("0000000000").shape == (123398, 4)
    assert load_frame("0000000153").shape == (98533, 4)


def test_frame0_defaults_match_survey_probe():
    """76 547 ground / 46 851 obstacle; reference clustering 572 clusters, 1 121 INVALID"""
    pts = load_frame("0000000000")
    r = oracle.segment(pts)
    assert (len(r["ground_idx"]), len(r["obstacle_idx"])) == (76547, 46851)
    lab, nc, nexp, _ = oracle.cluster(pts[r["obstacle_idx"]], stats=True)
    assert nc == 572 and int((lab == oracle.INVALID).sum()) == 1121 and nexp == 11552


@pytest.mark.parametrize("frame", FRAMES)
@pytest.mark.parametrize("sname", list(SEG_CFGS))
def test_segmentation_golden(frame, sname):
    pts = load_frame(frame)
    r = oracle.segment(pts, oracle.SegCfg(**SEG_CFGS[sname]))
    g = gold()
    assert np.array_equal(r["labels"].astype(np.uint8), g[f"seg_{frame}_{sname}_labels"])
    assert np.array_equal(r["planes"], g[f"seg_{frame}_{sname}_planes"])
    assert tuple(g[f"seg_{frame}_{sname}_counts"]) == (len(r["ground_idx"]), len(r["obstacle_idx"]))
    # Q7: output order is the x-sorted order filtered by label
    x = pts[:, 0]
    P = SEG_CFGS[sname]["number_of_planar_partitions"]
    n_per = pts.shape[0] // P
    order = np.lexsort((np.arange(pts.shape[0]), x))
    expect_o = [i for i in order[: n_per * P] if r["labels"][i] == oracle.OBSTACLE]
    assert np.array_equal(r["obstacle_idx"], np.array(expect_o, np.uint32))
    assert (r["labels"][order[n_per * P:]] == oracle.UNKNOWN).all()  # Q2


@pytest.mark.parametrize("frame", FRAMES)
@pytest.mark.parametrize("sname", ["p2i3", "p6i5"])
@pytest.mark.parametrize("cname", list(CLU_CFGS))
def test_clustering_golden_from_reference_build(frame, sname, cname):
    g = gold()
    key = f"clu_{frame}_{sname}_{cname}_labels"
    if key not in g:
        pytest.skip("not generated for this frame")
    pts = load_frame(frame)
    obs = pts[oracle.segment(pts, oracle.SegCfg(**SEG_CFGS[sname]))["obstacle_idx"]]
    d2, q = CLU_CFGS[cname]
    lab, nc = oracle.cluster(obs, oracle.CluCfg(d2, q))
    assert nc == int(g[f"clu_{frame}_{sname}_{cname}_n"][0])
    assert np.array_equal(lab, g[key])


def test_kd_preorder_golden():
    pts = load_frame("0000000000")
    obs = pts[oracle.segment(pts)["obstacle_idx"]]
    assert np.array_equal(oracle.kd_preorder(obs), gold()["kd_0000000000_p2i3_preorder"])


# ---- live against the reference's own headers (kdtree.hpp, queue.hpp) -------------------------------

def _cloud(kind, m, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return (rng.random((m, 3)) * 40 - 20).astype(np.float32)
    if kind == "ties":
        return (rng.integers(-40, 40, (m, 3)) * 0.05).astype(np.float32)
    if kind == "dups":
        base = (rng.random((max(m // 4, 1), 3)) * 10).astype(np.float32)
        return base[rng.integers(0, base.shape[0], m)]
    return np.full((m, 3), 1.5, np.float32)


@needs_ref
@pytest.mark.parametrize("kind", ["uniform", "ties", "dups", "const"])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 7, 16, 100, 1000, 20_000])
def test_kd_preorder_equals_reference_kdtree(kind, m):
    xyz = _cloud(kind, m, m)
    assert np.array_equal(oracle.kd_preorder(xyz), oracle.ref_kd_preorder(xyz))


@needs_ref
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 64, 1000, 4097])
def test_nth_element_equals_libstdcxx(n):
    """restated introselect == std::nth_element of this toolchain, including heavy ties"""
    rng = np.random.default_rng(n)
    for trial in range(20):
        keys = rng.integers(0, max(2, n // 3), n).astype(np.float32) if trial % 2 else rng.random(n).astype(np.float32)
        pay = np.arange(n, dtype=np.uint32)
        nth = int(rng.integers(0, n))
        k1, p1 = oracle.nth_element(keys, pay, 0, nth, n)
        k2, p2 = oracle.ref_nth_element(keys, pay, 0, nth, n)
        assert np.array_equal(k1, k2) and np.array_equal(p1, p2)


@needs_ref
def test_nth_element_depth_limit_fallback():
    """median-of-3 killer input drives introselect into its heap_select branch (bits/stl_algo.h:1970-1976)"""
    n = 2048
    keys = np.zeros(n, np.float32)
    # classic anti-quicksort arrangement for a median-of-(first+1, mid, last-1) pivot rule
    half = n // 2
    for i in range(half):
        keys[i] = i + 1 if i % 2 == 0 else half + i + (1 - i % 2)
    keys[half:] = np.arange(2, 2 * (n - half) + 2, 2)
    pay = np.arange(n, dtype=np.uint32)
    for nth in (1, n // 2, n - 2):
        k1, p1 = oracle.nth_element(keys, pay, 0, nth, n)
        k2, p2 = oracle.ref_nth_element(keys, pay, 0, nth, n)
        assert np.array_equal(p1, p2)


@needs_ref
@pytest.mark.parametrize("kind,r2", [("uniform", 4.0), ("ties", 0.04), ("dups", 0.3)])
def test_radius_search_equals_reference(kind, r2):
    xyz = _cloud(kind, 3000, 5)
    rng = np.random.default_rng(0)
    for j in rng.integers(0, 3000, 60):
        a, da = oracle.radius_search(xyz, xyz[j], r2)
        b, db = oracle.ref_radius_search(xyz, xyz[j], r2)
        assert np.array_equal(a, b) and np.array_equal(da.view(np.uint32), db.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("q", [0.0, 0.5, 1.0])
def test_fec_equals_reference_loop(q):
    pts = load_frame("0000000077")
    obs = pts[oracle.segment(pts)["obstacle_idx"]][:15000]
    a, na = oracle.cluster(obs, oracle.CluCfg(0.18, q))
    b, nb = oracle.ref_fec(obs, oracle.CluCfg(0.18, q))
    assert na == nb and np.array_equal(a, b)


# ---- mirrors the reference's own test (test/test_kdtree.cpp:97-187), float32, inclusive, seeded -----

def test_radius_search_matches_brute_force():
    rng = np.random.default_rng(1234)
    pts = (rng.random((1000, 3)) * 30 - 15).astype(np.float32)
    r2 = np.float32(4.0)
    for t in (rng.random((50, 3)) * 30 - 15).astype(np.float32):
        idx, dist = oracle.radius_search(pts, t, float(r2))
        d = t - pts
        bd = d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + (d[:, 2] * d[:, 2] + np.float32(0)))
        want = np.nonzero(bd <= r2)[0]
        assert sorted(idx.tolist()) == want.tolist()
        assert np.array_equal(np.sort(dist), np.sort(bd[want]))


# ---- known-answer tests ------------------------------------------------------------------------------

@pytest.mark.parametrize("a,b,c", [(0.0, 0.0, -1.7), (0.02, -0.01, -1.7), (-0.05, 0.03, 0.4)])
def test_plane_known_answer(a, b, c):
    """z = a x + b y + c + noise: normal within 1e-4 of the analytic one and pointing up (c > 0)"""
    rng = np.random.default_rng(7)
    n = 20000
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 0] = rng.random(n) * 100 - 50
    xyz[:, 1] = rng.random(n) * 100 - 50
    xyz[:, 2] = a * xyz[:, 0] + b * xyz[:, 1] + c + rng.normal(0, 0.01, n)
    plane, rc = oracle.plane_from_points(xyz)
    assert rc == 0
    nrm = np.array([-a, -b, 1.0]) / np.sqrt(a * a + b * b + 1.0)
    assert plane[2] > 0
    assert np.abs(plane[:3] - nrm).max() < 1e-4
    assert abs(plane[3] - c * nrm[2]) < 2e-3
    # float64 cross-check of the whole fit
    p = xyz.astype(np.float64)
    w, v = np.linalg.eigh(np.cov(p.T))
    n64 = v[:, 0] * np.sign(v[2, 0])
    assert np.abs(plane[:3] - n64).max() < 1e-4


def test_jacobi_svd_matches_numpy_and_is_orthonormal():
    rng = np.random.default_rng(3)
    for _ in range(200):
        m = rng.normal(size=(3, 3))
        cov = (m @ m.T).astype(np.float32)
        v, s = oracle.jacobi_svd3(cov)
        assert np.abs(v.T @ v - np.eye(3)).max() < 1e-5
        assert s[0] >= s[1] >= s[2] >= 0
        w = np.linalg.eigvalsh(cov.astype(np.float64))[::-1]
        assert np.allclose(s, w, rtol=1e-4, atol=1e-5 * w[0])
    v, s = oracle.jacobi_svd3(np.zeros((3, 3), np.float32))
    assert np.array_equal(v, np.eye(3, dtype=np.float32)) and (s == 0).all()


def test_plane_degenerate_inputs():
    assert oracle.plane_from_points(np.zeros((2, 3), np.float32))[1] == 1          # < 3 points
    plane, rc = oracle.plane_from_points(np.ones((10, 3), np.float32))             # identical points
    assert rc == 0 and tuple(plane[:3]) == (0.0, 0.0, 1.0)
    bad = np.zeros((4, 3), np.float32)
    bad[0, 0] = np.inf
    assert oracle.plane_from_points(bad)[1] == oracle.ERR_RANGE
    bad[0, 0] = np.nan
    assert oracle.plane_from_points(bad)[1] == oracle.ERR_RANGE


def test_plane_far_coordinates_are_exact():
    """any finite coordinate is accepted: the integer moments are exact up to 2^24 m (UTM / ECEF scale), so a
    cloud shifted by a multiple of a large power of two keeps its normal (float64 cross-check)"""
    rng = np.random.default_rng(11)
    n = 5000
    xyz = np.zeros((n, 3), np.float64)
    xyz[:, 0] = np.round(rng.random(n) * 64, 0)
    xyz[:, 1] = np.round(rng.random(n) * 64, 0)
    xyz[:, 2] = np.round(0.25 * xyz[:, 0] - 0.5 * xyz[:, 1] + rng.normal(0, 1.0, n), 0)
    for shift in (0.0, 4096.0, 500_000.0, 8_000_000.0):
        p = (xyz + [shift, -shift, shift / 2]).astype(np.float32)
        assert np.array_equal(p.astype(np.float64), xyz + [shift, -shift, shift / 2])  # exactly representable
        plane, rc = oracle.plane_from_points(p)
        assert rc == 0
        w, v = np.linalg.eigh(np.cov(xyz.T))
        nrm = v[:, 0] * np.sign(v[2, 0])
        assert np.abs(plane[:3] * np.sign(plane[2]) - nrm).max() < 2e-5, (shift, plane, nrm)
    # beyond 2^24 m the moments use the clamped coordinate: still a result, never an error
    p = xyz.astype(np.float32)
    p[0] = [3.0e38, -3.0e38, 1.0e30]
    assert oracle.plane_from_points(p)[1] == 0


def test_segment_edge_cases():
    assert oracle.segment(np.zeros((0, 4), np.float32))["labels"].shape == (0,)
    # N mod P != 0: the highest-x leftovers stay UNKNOWN (Q2)
    rng = np.random.default_rng(2)
    pts = np.zeros((1001, 4), np.float32)
    pts[:, 0] = rng.random(1001) * 40
    pts[:, 1] = rng.random(1001) * 40
    pts[:, 2] = -1.7 + rng.normal(0, 0.03, 1001)
    pts[:300, 2] += 1.0
    r = oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6))
    assert 1001 % 6 == 5 and (r["labels"] == oracle.UNKNOWN).sum() == 5
    assert set(np.nonzero(r["labels"] == 0)[0]) == set(np.argsort(pts[:, 0], kind="stable")[-5:])
    # fewer than 3 points per segment: nothing labelled (Q5)
    r = oracle.segment(pts[:5], oracle.SegCfg(number_of_planar_partitions=2))
    assert (r["labels"] == 0).all() and (r["status"] == oracle.SEG_TOO_FEW_POINTS).all()
    # no point above mean + seed threshold: no seeds -> all obstacle (Q4)
    flat = pts.copy()
    flat[:, 2] = -1.7 + rng.random(1001) * 0.1
    r = oracle.segment(flat, oracle.SegCfg(number_of_planar_partitions=1))
    assert (r["labels"] == oracle.OBSTACLE).all() and r["status"][0] == oracle.SEG_ALL_OBSTACLE
    # signed inlier test: everything below the plane is ground (Q1)
    pts2 = pts.copy()
    pts2[-50:, 2] = -2.3
    pts2[-50:, 0] = rng.random(50) * 39  # keep them inside the segment
    r = oracle.segment(pts2, oracle.SegCfg(number_of_planar_partitions=1))
    assert (r["labels"][-50:] == oracle.GROUND).all()


def test_cluster_edge_cases():
    lab, nc = oracle.cluster(np.zeros((0, 3), np.float32))
    assert lab.shape == (0,) and nc == 0
    # 2-point component: 3 touches < 4 -> INVALID; 3-point component: >= 5 touches -> valid (Q8)
    p = np.zeros((5, 3), np.float32)
    p[:, 0] = [0, 0.3, 5, 5.3, 5.6]
    lab, nc = oracle.cluster(p, oracle.CluCfg(0.18, 1.0, 4))
    assert lab.tolist() == [-1, -1, 0, 0, 0] and nc == 1
    # two points exactly at distance d are neighbours (inclusive <=, src/kdtree.hpp:315)
    p = np.zeros((2, 3), np.float32)
    p[1, 0] = 0.5
    assert oracle.cluster(p, oracle.CluCfg(0.25, 1.0, 1))[0].tolist() == [0, 0]
    p[1, 0] = np.nextafter(np.float32(0.5), np.float32(1))
    assert oracle.cluster(p, oracle.CluCfg(0.25, 1.0, 1))[0].tolist() == [0, 1]
    # quality 0: each cluster is one ball around its seed (Q9); duplicates are absorbed at dist 0
    p = np.zeros((4, 3), np.float32)
    p[:, 0] = [0, 0.4, 0.8, 0.8]
    assert oracle.cluster(p, oracle.CluCfg(0.25, 0.0, 1))[0].tolist() == [0, 0, 1, 1]
    assert oracle.cluster(p, oracle.CluCfg(0.25, 1.0, 1))[0].tolist() == [0, 0, 0, 0]


def test_quality_one_is_connected_components():
    """q = 1: partition == connected components of the d-graph"""
    rng = np.random.default_rng(11)
    p = np.round(rng.random((3000, 3)) * [30, 30, 1], 2).astype(np.float32)
    lab, nc = oracle.cluster(p, oracle.CluCfg(0.18, 1.0, 1))
    root = brute_components(p, 0.18)
    assert np.array_equal(partition_signature(lab), root.astype(np.int64))


def test_synthetic_scene_is_deterministic():
    a = synthetic_scene(6000, 20, 200, 20240601)
    b = synthetic_scene(6000, 20, 200, 20240601)
    assert a.shape == (10000, 4) and np.array_equal(a, b)
    assert np.array_equal(a[:, :3], np.round(a[:, :3].astype(np.float64), 3).astype(np.float32))


# ---- N3: convex hull restatement (Andrew monotone chain; the reference's Convex-Hull submodule is absent) ----

def test_convex_hull_known_answers():
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5], [0.5, 0], [1, 0.5], [0, 0], [1, 1]], np.float32)
    assert oracle.convex_hull(sq).tolist() == [0, 1, 2, 3]                      # corners, CCW, first duplicate kept
    assert oracle.convex_hull(np.array([[0, 0], [1, 1], [2, 2], [3, 3]], np.float32)).tolist() == [0, 3]  # collinear
    assert oracle.convex_hull(np.array([[1, 1], [1, 1]], np.float32)).tolist() == [0]
    assert oracle.convex_hull(np.zeros((0, 2), np.float32)).tolist() == []
    assert oracle.convex_hull(np.array([[3, 4]], np.float32)).tolist() == [0]
    assert oracle.convex_hull(np.array([[0, 0], [4, 0], [2, -3]], np.float32)).tolist() == [0, 2, 1]  # CCW
    assert oracle.convex_hull(np.array([[-0.0, 0.0], [0.0, -0.0], [1, 0], [0, 1]], np.float32)).tolist() == [0, 2, 3]


@pytest.mark.parametrize("seed", range(5))
def test_convex_hull_against_scipy(seed):
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    for _ in range(60):
        n = int(rng.integers(3, 200))
        p = rng.normal(size=(n, 2)).astype(np.float32)
        h = oracle.convex_hull(p)
        assert set(h.tolist()) == set(ConvexHull(p.astype(np.float64)).vertices.tolist())
        x, y = p[h, 0].astype(np.float64), p[h, 1].astype(np.float64)
        assert 0.5 * np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y) > 0          # counter-clockwise
        assert h[0] == np.lexsort((p[:, 1], p[:, 0]))[0]                           # starts at the lowest (x, y)


def test_cluster_hulls_follow_the_reference_size_rule():
    """src/polygon_simplification.cpp:97: only clusters with fewer than 20 points take the convex branch"""
    pts = load_frame(FRAMES[0])
    obs = pts[oracle.segment(pts)["obstacle_idx"]]
    lab, nc = oracle.cluster(obs)
    off, idx = oracle.cluster_hulls(obs, lab, nc, 20)
    sizes = np.bincount(lab[lab >= 0], minlength=nc)
    assert ((np.diff(off) > 0) == (sizes < 20)).all() and len(idx) == off[-1]
    c = int(np.nonzero(sizes < 20)[0][0])
    mem = np.nonzero(lab == c)[0]
    assert np.array_equal(idx[off[c]:off[c + 1]], mem[oracle.convex_hull(obs[mem, :2])])
