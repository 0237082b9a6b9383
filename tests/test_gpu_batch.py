"""GPU parity tests of the device-resident entry points and of the multi-frame launch chain
(lpx_segment_cluster_device, lpx_segment_cluster_batch_device), through the C-ABI.

Bar: every frame of a batch gets exactly what the single-frame path (itself checked against the oracle and
the reference goldens in test_gpu_pipeline.py) returns for that frame: labels, index lists, plane words,
cluster labels and counts, bit for bit; the oracle is compared directly as well."""
import numpy as np
import pytest

import oracle
from lidar_processing_amd import ClusteringConfiguration, Context, LpxError, SegmentationConfiguration
from util import FRAMES, load_frame, synthetic_scene

pytestmark = pytest.mark.gpu

SEG = dict(number_of_planar_partitions=6, number_of_iterations=5)
CLU = dict(distance_squared=0.25, cluster_quality=0.5)


def run_batch(bctx, clouds, seg_kw, clu_kw, stride_floats=4):
    """clouds: list of (n_i, >=3) float arrays -> list of per-frame result dicts (device API, pitched arrays)"""
    import torch
    dev = torch.device("cuda:0")
    B = len(clouds)
    pitch = max(1, max(c.shape[0] for c in clouds)) + 7
    P = seg_kw.get("number_of_planar_partitions", 2)
    host = np.zeros((B, pitch, stride_floats), np.float32)
    for b, c in enumerate(clouds):
        host[b, :c.shape[0], :min(stride_floats, c.shape[1])] = c[:, :stride_floats]
    d_pts = torch.from_numpy(host).to(dev)
    d_labels = torch.full((B, pitch), 0xdeadbeef, dtype=torch.int64, device=dev).to(torch.int32)
    d_gidx = torch.zeros((B, pitch), dtype=torch.int32, device=dev)
    d_oidx = torch.zeros((B, pitch), dtype=torch.int32, device=dev)
    d_planes = torch.full((B, 4 * P), 7.0, dtype=torch.float32, device=dev)
    d_clab = torch.full((B, pitch), -77, dtype=torch.int32, device=dev)
    d_counts = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    n = [c.shape[0] for c in clouds]
    bctx.segment_cluster_batch_device(n, d_pts.data_ptr(), 4 * stride_floats, pitch, SegmentationConfiguration(**seg_kw),
                                      ClusteringConfiguration(**clu_kw), d_labels.data_ptr(), d_gidx.data_ptr(),
                                      d_oidx.data_ptr(), d_planes.data_ptr(), d_clab.data_ptr(), d_counts.data_ptr())
    bctx.synchronize()
    counts = d_counts.cpu().numpy().view(np.uint32)
    out = []
    for b in range(B):
        ng, no, nc, status = (int(v) for v in counts[b])
        out.append(dict(status=status, n_clusters=nc,
                        labels=d_labels[b, :n[b]].cpu().numpy().view(np.uint32),
                        ground_idx=d_gidx[b, :ng].cpu().numpy().view(np.uint32),
                        obstacle_idx=d_oidx[b, :no].cpu().numpy().view(np.uint32),
                        planes=d_planes[b].cpu().numpy().reshape(P, 4),
                        cluster_labels=d_clab[b, :no].cpu().numpy(),
                        tail_labels=d_clab[b, no:].cpu().numpy()))
    return out


def check_frame(res, ref):
    assert res["status"] == 0
    assert np.array_equal(res["labels"], ref["labels"])
    assert np.array_equal(res["ground_idx"], ref["ground_idx"])
    assert np.array_equal(res["obstacle_idx"], ref["obstacle_idx"])
    assert np.array_equal(res["planes"].view(np.uint32), ref["planes"].view(np.uint32))
    if not np.array_equal(res["cluster_labels"], ref["cluster_labels"]):
        d = np.nonzero(res["cluster_labels"] != ref["cluster_labels"])[0]
        raise AssertionError(f"cluster labels differ at {len(d)} of {len(ref['cluster_labels'])} points, first {d[:6]}: "
                             f"{res['cluster_labels'][d[:6]]} against {ref['cluster_labels'][d[:6]]}; clusters "
                             f"{res['n_clusters']} against {ref['n_clusters']}")
    assert res["n_clusters"] == ref["n_clusters"]
    assert (res["tail_labels"] == -77).all(), "wrote past the obstacle count of the frame"


def single(ctx, cloud, seg_kw, clu_kw):
    return ctx.segment_cluster(cloud, SegmentationConfiguration(**seg_kw), ClusteringConfiguration(**clu_kw))


def test_batch_of_real_frames_matches_single_frame_path_and_oracle(ctx):
    clouds = [load_frame(f) for f in FRAMES] + [load_frame(FRAMES[0])[:50_000]]
    bctx = Context(0, batch=4)
    try:
        res = run_batch(bctx, clouds, SEG, CLU)
    finally:
        bctx.close()
    for c, r in zip(clouds, res):
        check_frame(r, single(ctx, c, SEG, CLU))
        o = oracle.segment(c, oracle.SegCfg(**SEG))
        assert np.array_equal(r["labels"], o["labels"]) and np.array_equal(r["obstacle_idx"], o["obstacle_idx"])
        oc_labels = oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**CLU))[0]
        assert np.array_equal(r["cluster_labels"], oc_labels)


@pytest.mark.parametrize("B", [1, 2, 7])
def test_batch_ragged_frames(ctx, B):
    """frames of very different sizes in one call, including an empty one and one with fewer points than
    partitions (every label UNKNOWN, src/segmentation.cpp:104-149)"""
    sizes = [30_000, 0, 3, 11_111, 4096 * 2 + 1, 64, 20_001][:B]
    clouds = []
    for i, n in enumerate(sizes):
        if n >= 64:
            clouds.append(synthetic_scene(n - n // 3, 8, max(1, (n // 3) // 8), seed=100 + i)[:n])
        else:
            clouds.append(synthetic_scene(64, 1, 8, seed=100 + i)[:n])
    seg_kw = dict(number_of_planar_partitions=4, number_of_iterations=3)
    clu_kw = dict(distance_squared=0.36, cluster_quality=0.3, min_cluster_size=3)
    bctx = Context(0, batch=8)
    try:
        res = run_batch(bctx, clouds, seg_kw, clu_kw)
        res2 = run_batch(bctx, clouds[::-1], seg_kw, clu_kw)[::-1]  # slots are reused with other sizes
    finally:
        bctx.close()
    for c, r, r2 in zip(clouds, res, res2):
        ref = single(ctx, c, seg_kw, clu_kw)
        check_frame(r, ref)
        check_frame(r2, ref)


def test_batch_large_segments_take_the_pass_per_launch_path(ctx):
    """segments above the single-workgroup limit (24576 points) in a batch"""
    clouds = [synthetic_scene(120_000, 60, 500, seed=7), synthetic_scene(90_000, 40, 400, seed=8)]
    seg_kw = dict(number_of_planar_partitions=2, number_of_iterations=3)
    bctx = Context(0, batch=2)
    try:
        res = run_batch(bctx, clouds, seg_kw, CLU)
    finally:
        bctx.close()
    for c, r in zip(clouds, res):
        check_frame(r, single(ctx, c, seg_kw, CLU))


@pytest.mark.parametrize("mode", ["search", "lists"])
def test_component_sort_passes_above_the_obstacle_count_are_per_frame(ctx, mode):
    """The host sizes the component sort for its bound of the obstacle count (here 17-18 bits); a frame with at most
    2^16 obstacle points takes the identity form of the third pass, a frame with more the real one -- side by side in one
    launch (csrc/lpx_primitives.hip: SORT_KEYS_BELOW_N, SORT_PREFIXED), and a frame far below the bound leaves most of
    the launch's tiles empty."""
    clouds = [synthetic_scene(40_000, 300, 300, seed=31),     # ~90k obstacle points: above 2^16
              synthetic_scene(90_000, 100, 300, seed=32),     # ~30k: identity third pass
              synthetic_scene(3_000, 10, 100, seed=33),       # ~1k: one tile of sixty
              synthetic_scene(50_000, 220, 300, seed=34)]     # ~66k: around the boundary
    seg_kw = dict(number_of_planar_partitions=4, number_of_iterations=3)
    clu_kw = dict(distance_squared=0.16, cluster_quality=0.5)
    bctx = Context(0, batch=4)
    try:
        bctx.set_neighbour_mode(mode)
        res = run_batch(bctx, clouds, seg_kw, clu_kw)
    finally:
        bctx.close()
    counts = [len(r["obstacle_idx"]) for r in res]
    assert max(counts) > 65536 > min(counts), counts
    for c, r in zip(clouds, res):
        check_frame(r, single(ctx, c, seg_kw, clu_kw))


def test_batch_stride_and_far_returns_are_per_frame(ctx):
    good = load_frame(FRAMES[1])[:20_000]
    far = good.copy()
    far[123, 1] = 5000.0  # beyond the int32 fixed-point range: that frame takes the wide-moment path, alone
    far[123, 2] = -1.7
    bad = good.copy()
    bad[5, 0] = np.nan
    bctx = Context(0, batch=4)
    try:
        res = run_batch(bctx, [good, far, bad, good], SEG, CLU, stride_floats=8)
    finally:
        bctx.close()
    ref = single(ctx, good, SEG, CLU)
    check_frame(res[0], ref)
    check_frame(res[3], ref)
    check_frame(res[1], single(ctx, far, SEG, CLU))
    o = oracle.segment(far, oracle.SegCfg(**SEG))
    assert o["rc"] == 0 and np.array_equal(res[1]["labels"], o["labels"])
    assert np.array_equal(res[1]["planes"].view(np.uint32), o["planes"].view(np.uint32))
    assert res[2]["status"] == 2  # -LPX_ERR_RANGE: NaN


def test_batch_non_finite_frame_is_flagged_and_not_clustered(ctx):
    good = load_frame(FRAMES[0])[:30_000]
    bad = good.copy()
    bad[7, 0] = np.nan
    bad[20_000, 2] = np.inf
    bad[29_999, 1] = -np.inf
    bctx = Context(0, batch=3)
    try:
        res = run_batch(bctx, [bad, good, bad], SEG, CLU)
    finally:
        bctx.close()
    check_frame(res[1], single(ctx, good, SEG, CLU))
    for r in (res[0], res[2]):
        assert r["status"] == 2  # -LPX_ERR_RANGE
        assert r["n_clusters"] == 0 and r["obstacle_idx"].shape[0] == 0  # nothing was handed to the clustering
    with pytest.raises(LpxError):
        single(ctx, bad, SEG, CLU)
    with pytest.raises(LpxError):
        ctx.cluster(bad, ClusteringConfiguration(**CLU))
    assert single(ctx, good, SEG, CLU)["n_clusters"] > 0  # the context is fine afterwards


def test_batch_argument_errors(ctx):
    bctx = Context(0, batch=2)
    try:
        with pytest.raises(LpxError):
            run_batch(bctx, [load_frame(FRAMES[0])[:1000]] * 3, SEG, CLU)  # more frames than slots
        res = run_batch(bctx, [load_frame(FRAMES[0])[:1000]] * 2, SEG, CLU)  # and the context still works
        assert res[0]["status"] == 0
    finally:
        bctx.close()
    with pytest.raises(LpxError):
        Context(0, batch=65)


def test_batch_context_serves_single_frame_entry_points(ctx):
    c = load_frame(FRAMES[2])
    bctx = Context(0, batch=3)
    try:
        a = single(bctx, c, SEG, CLU)
    finally:
        bctx.close()
    b = single(ctx, c, SEG, CLU)
    for k in ("labels", "ground_idx", "obstacle_idx", "cluster_labels"):
        assert np.array_equal(a[k], b[k])


def test_device_entry_point_matches_host_entry_point(ctx):
    import torch
    dev = torch.device("cuda:0")
    c = load_frame(FRAMES[0])
    n = c.shape[0]
    d_pts = torch.from_numpy(c).to(dev)
    d_labels = torch.zeros(n, dtype=torch.int32, device=dev)
    d_gidx = torch.zeros(n, dtype=torch.int32, device=dev)
    d_oidx = torch.zeros(n, dtype=torch.int32, device=dev)
    d_planes = torch.zeros(4 * 6, dtype=torch.float32, device=dev)
    d_clab = torch.zeros(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.segment_cluster_device(d_pts.data_ptr(), 16, n, SegmentationConfiguration(**SEG), ClusteringConfiguration(**CLU),
                               d_labels.data_ptr(), d_gidx.data_ptr(), d_oidx.data_ptr(), d_planes.data_ptr(),
                               d_clab.data_ptr(), d_counts.data_ptr())
    ctx.synchronize()
    ng, no, nc, status = (int(v) for v in d_counts.cpu().numpy().view(np.uint32))
    ref = single(ctx, c, SEG, CLU)
    assert status == 0 and nc == ref["n_clusters"]
    assert np.array_equal(d_labels.cpu().numpy().view(np.uint32), ref["labels"])
    assert np.array_equal(d_gidx[:ng].cpu().numpy().view(np.uint32), ref["ground_idx"])
    assert np.array_equal(d_oidx[:no].cpu().numpy().view(np.uint32), ref["obstacle_idx"])
    assert np.array_equal(d_clab[:no].cpu().numpy(), ref["cluster_labels"])
    assert np.array_equal(d_planes.cpu().numpy().view(np.uint32).reshape(6, 4), ref["planes"].view(np.uint32))


@pytest.mark.parametrize("byte", ["0xa5", "0xff"])
def test_results_do_not_depend_on_what_an_earlier_frame_left_in_the_workspace(byte):
    """LPX_POISON (development build, liblpx_dev.so) fills the per-point workspace and the neighbour lists before every new frame (read once per
    process, hence the subprocess): the batch tests above and the capacity-retry tests must pass unchanged, and
    a frame whose lists overflow must not chase the poison (it used to fault before the retry could run)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    from lidar_processing_amd import _lib
    env = dict(os.environ, LPX_POISON=byte, LPX_LIB=_lib.DEV_LIB_PATH)  # the knob exists in the development build only
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_batch.py"), os.path.join(here, "test_gpu_pipeline.py"),
                        "-k", "not earlier_frame and (test_batch or workspace_grows or workspace_retry or real_frames "
                              "or edge_cases)"],  # (never this test itself: it would recurse)
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_randomised_scenes_and_configurations_match_the_oracle():
    """tools/fuzz.py for a quarter of a minute: random scenes x random configurations through the host path (both
    neighbour modes) and through 3-frame chains on long-lived contexts, everything compared with the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "15", "11"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", ["search", "lists"])
def test_batch_of_large_frames_takes_the_multi_workgroup_kd_rounds(mode):
    """two frames of 1M and 0.7M points in ONE chain: obstacle clouds above 131 072 points send the top kd levels of
    BOTH slots through kd_top_* (per-slot introselect state and tile counts), the radix tables through
    hist_rows_kernel and the label scan through the tile-sum path; everything equals the oracle"""
    clouds = [synthetic_scene(600_000, 2000, 200, 20240601), synthetic_scene(400_000, 1500, 200, 777)]
    seg_kw = dict(number_of_planar_partitions=12, number_of_iterations=3)
    clu_kw = dict(distance_squared=0.09, cluster_quality=0.5)
    bctx = Context(0, batch=2)
    bctx.set_neighbour_mode(mode)
    try:
        res = run_batch(bctx, clouds, seg_kw, clu_kw)
        res2 = run_batch(bctx, clouds[::-1], seg_kw, clu_kw)[::-1]  # slots swapped
    finally:
        bctx.close()
    for c, r, r2 in zip(clouds, res, res2):
        o = oracle.segment(c, oracle.SegCfg(**seg_kw))
        lab, nc = oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**clu_kw))
        assert o["obstacle_idx"].shape[0] > 131_072
        for rr in (r, r2):
            assert rr["status"] == 0
            assert np.array_equal(rr["labels"], o["labels"]) and np.array_equal(rr["obstacle_idx"], o["obstacle_idx"])
            assert np.array_equal(rr["ground_idx"], o["ground_idx"])
            assert np.array_equal(rr["planes"].view(np.uint32), o["planes"].view(np.uint32))
            assert np.array_equal(rr["cluster_labels"], lab) and rr["n_clusters"] == nc


def _enqueue_only(bctx, clouds, seg_kw, clu_kw):
    """like run_batch without the synchronisation: returns the device arrays and a reader"""
    import torch
    dev = torch.device("cuda:0")
    B = len(clouds)
    pitch = max(1, max(c.shape[0] for c in clouds)) + 7
    P = seg_kw.get("number_of_planar_partitions", 2)
    host = np.zeros((B, pitch, 4), np.float32)
    for b, c in enumerate(clouds):
        host[b, :c.shape[0], :4] = c[:, :4]
    d = dict(pts=torch.from_numpy(host).to(dev),
             labels=torch.full((B, pitch), 0x7eadbeef, dtype=torch.int32, device=dev),
             gidx=torch.zeros((B, pitch), dtype=torch.int32, device=dev),
             oidx=torch.zeros((B, pitch), dtype=torch.int32, device=dev),
             planes=torch.full((B, 4 * P), 7.0, dtype=torch.float32, device=dev),
             clab=torch.full((B, pitch), -77, dtype=torch.int32, device=dev),
             counts=torch.zeros((B, 4), dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    n = [c.shape[0] for c in clouds]
    bctx.segment_cluster_batch_device(n, d["pts"].data_ptr(), 16, pitch, SegmentationConfiguration(**seg_kw),
                                      ClusteringConfiguration(**clu_kw), d["labels"].data_ptr(), d["gidx"].data_ptr(),
                                      d["oidx"].data_ptr(), d["planes"].data_ptr(), d["clab"].data_ptr(),
                                      d["counts"].data_ptr())

    def read():
        counts = d["counts"].cpu().numpy().view(np.uint32)
        out = []
        for b in range(B):
            ng, no, nc, status = (int(v) for v in counts[b])
            out.append(dict(status=status, n_clusters=nc, labels=d["labels"][b, :n[b]].cpu().numpy().view(np.uint32),
                            ground_idx=d["gidx"][b, :ng].cpu().numpy().view(np.uint32),
                            obstacle_idx=d["oidx"][b, :no].cpu().numpy().view(np.uint32),
                            planes=d["planes"][b].cpu().numpy().reshape(P, 4),
                            cluster_labels=d["clab"][b, :no].cpu().numpy(), tail_labels=d["clab"][b, no:].cpu().numpy()))
        return out
    return read


@pytest.mark.parametrize("mode", ["search", "lists"])
def test_overlapped_tail_gives_the_same_results(ctx, mode):
    """lpx_set_overlap: five back-to-back calls alternate between the two slot sets, the replay and label kernels of
    each run on the tail stream beside the front end of the next call, nothing is read before ONE synchronize at the
    end -- every frame of every call must equal the single-frame path; then the context is switched back, and the
    statistics / coloured clouds of the last call come from the slot set that served it"""
    import torch
    frames = [load_frame(f) for f in FRAMES]
    rng = np.random.default_rng(7)
    calls = []
    for k in range(5):
        calls.append([frames[(k + j) % 3][rng.integers(0, 3)::(2 + (k + j) % 3)].copy() for j in range(4)])
    refs = {}
    for k, clouds in enumerate(calls):
        for j, c in enumerate(clouds):
            refs[(k, j)] = single(ctx, c, SEG, CLU)
    bctx = Context(0, batch=4)
    try:
        bctx.set_neighbour_mode(mode)
        bctx.set_overlap(True)
        readers = [_enqueue_only(bctx, clouds, SEG, CLU) for clouds in calls]  # no synchronisation in between
        bctx.synchronize()
        for k, read in enumerate(readers):
            for j, res in enumerate(read()):
                check_frame(res, refs[(k, j)])
        # the statistics of the last call (served by the primary slot set: call 4 is even)
        st = bctx.frame_stats(0)
        assert st["n_obstacle"] == len(refs[(4, 0)]["obstacle_idx"]) and st["n_clusters"] == refs[(4, 0)]["n_clusters"]
        # an odd number of further calls ends on the twin
        read = _enqueue_only(bctx, calls[1], SEG, CLU)
        st = bctx.frame_stats(1)  # synchronises, reads the twin
        assert st["n_obstacle"] == len(refs[(1, 1)]["obstacle_idx"])
        for j, res in enumerate(read()):
            check_frame(res, refs[(1, j)])
        # lpx_wait_previous: a pipelined caller enqueues call k, then collects call k - 1 while k is still in flight
        prev = None
        for k in (0, 3, 1, 4):
            cur = _enqueue_only(bctx, calls[k], SEG, CLU)
            bctx.wait_previous()
            if prev is not None:
                for j, res in enumerate(prev[1]()):
                    check_frame(res, refs[(prev[0], j)])
            prev = (k, cur)
        bctx.synchronize()
        for j, res in enumerate(prev[1]()):
            check_frame(res, refs[(prev[0], j)])
        bctx.set_overlap(False)
        for j, res in enumerate(run_batch(bctx, calls[2], SEG, CLU)):
            check_frame(res, refs[(2, j)])
        bctx.set_overlap(True)  # and on again: the twin is kept
        for k in (3, 0):
            for j, res in enumerate(run_batch(bctx, calls[k], SEG, CLU)):
                check_frame(res, refs[(k, j)])
    finally:
        bctx.close()


def test_overlap_is_for_batch_contexts(ctx):
    with pytest.raises(LpxError):
        ctx.set_overlap(True)


@pytest.mark.parametrize("mask", ["ff", "bf", "40", "0"])
def test_xcd_affine_launch_geometry_is_a_bijection(mask):
    """lpx_block re-reads the linear workgroup number so that a frame's workgroups share an XCD; which kernel families do
    so is chosen per launch (LPX_REMAP, read once per process -- hence the subprocess).  Every family on, the library's
    default for small frames, only the family the default leaves out, and none: ragged batches of 8, 13, 16 and 19
    frames (partial groups of eight, empty frames, frames with fewer points than partitions) in both neighbour modes
    must equal the single-frame path, which is never re-read"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    from lidar_processing_amd import _lib
    env = dict(os.environ, LPX_REMAP=mask, LPX_LIB=_lib.DEV_LIB_PATH, PYTHONPATH=os.pathsep.join([os.path.dirname(here), here,
                                                                       os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(here, "remap_check.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "remap check ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("method", ["chunks", "grid", "sweep", "sweep_plain"])
def test_both_component_searches_give_the_single_frame_results(method):
    """the search path finds the connected components of the d-graph either from the kd groups' chunk tables (the
    default for frames of 400k points and more) or from the clique-cell grid (smaller frames); LPX_CC forces one --
    development library, read once per process, hence the subprocess.  Ragged batches in both neighbour modes against
    the single-frame path (tests/remap_check.py)"""
    import os
    import subprocess
    import sys
    from lidar_processing_amd import _lib
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, LPX_CC=method.split("_")[0], LPX_SWEEP_PLAIN="1" if method.endswith("_plain") else "0",
               LPX_LIB=_lib.DEV_LIB_PATH,
               PYTHONPATH=os.pathsep.join([os.path.dirname(here), here, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(here, "remap_check.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "remap check ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_ragged_batch_with_long_segments(ctx):
    """one frame whose segments exceed a workgroup's registers puts the whole chain on the many-workgroup seed selection:
    the short frames beside it (down to fewer points than partitions) must still equal their single-frame results"""
    big = np.concatenate([load_frame(f) for f in FRAMES])[:200_000]
    clouds = [big, load_frame(FRAMES[0])[:1], load_frame(FRAMES[1])[:100], load_frame(FRAMES[2])[:30_000], big[:60_001]]
    seg = dict(number_of_planar_partitions=2, number_of_iterations=3)
    for mode in ("search", "lists"):
        bctx = Context(0, batch=len(clouds))
        try:
            bctx.set_neighbour_mode(mode)
            res = run_batch(bctx, clouds, seg, CLU)
        finally:
            bctx.close()
        for j, (c, r) in enumerate(zip(clouds, res)):
            ref = single(ctx, c, seg, CLU)
            try:
                check_frame(r, ref)
            except AssertionError as e:
                o = oracle.segment(c, oracle.SegCfg(**seg))
                want = oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**CLU))[0] if c.shape[0] else np.zeros(0, np.int32)
                raise AssertionError(f"batch mode {mode}, cloud {j}: {e}; batch == oracle: "
                                     f"{np.array_equal(r['cluster_labels'], want)}, single == oracle: "
                                     f"{np.array_equal(ref['cluster_labels'], want)}") from e


def test_forked_front_end_gives_the_same_results():
    """lpx_set_fork: the component grid of a chain on a side stream beside its kd build and chunk tables -- alone and
    together with lpx_set_overlap, several calls back to back on one context"""
    clouds = [load_frame(f)[:30_000 + 3000 * i] for i, f in enumerate(FRAMES * 3)]
    refs = None
    one = Context(0)
    try:
        refs = [single(one, c, SEG, CLU) for c in clouds]
    finally:
        one.close()
    for overlap in (False, True):
        bctx = Context(0, batch=len(clouds))
        try:
            if overlap:
                bctx.set_overlap(True)
            bctx.set_fork(True)
            for _ in range(3):
                for res, ref in zip(run_batch(bctx, clouds, SEG, CLU), refs):
                    check_frame(res, ref)
        finally:
            bctx.close()


def test_first_call_of_a_fresh_list_context_is_deterministic():
    """A latent race of rounds 1-5, found in round 6 (tools/r6_flaky.py): near the end of a sub-region of the single-pass
    list workspace the four wavefronts of a bucket group could disagree on whether the group reserves by upper bounds or
    counts exact lengths (every thread read the moving cursor for itself) and the group's lists came out wrong -- 2-5
    partitions in 1000 differed once the region was 192 words per point.  Only the FIRST call of a context can be near
    the end (the workspace then grows on the evidence), so: 150 fresh batch contexts in lists mode on the dense 200k-point
    cloud, every partition equal to the oracle's."""
    big = np.concatenate([load_frame(f) for f in FRAMES])[:200_000]
    clouds = [big, load_frame(FRAMES[2])[:30_000]]
    seg = dict(number_of_planar_partitions=2, number_of_iterations=3)
    want = []
    for c in clouds:
        o = oracle.segment(c, oracle.SegCfg(**seg))
        want.append(oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**CLU))[0])
    bad = []
    for rep in range(150):
        b = Context(0, batch=len(clouds))
        try:
            b.set_neighbour_mode("lists")
            for j, r in enumerate(run_batch(b, clouds, seg, CLU)):
                if r["status"] != 0 or not np.array_equal(r["cluster_labels"], want[j]):
                    bad.append((rep, j, r["status"]))
        finally:
            b.close()
    assert not bad, bad[:5]
