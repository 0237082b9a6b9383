"""N1: binary PCD files -> pinned host memory -> double-buffered H2D / launch chain / D2H (lpx_pcd_load,
lpx_feeder_*).  The GPU box has no reference checkout, so the .pcd files are written here from the committed
stream fixture (bit-identical payloads, with the trailing bytes the reference's files carry)."""
import numpy as np
import pytest

from lidar_processing_amd import (ClusteringConfiguration, Context, Feeder, LpxError, SegmentationConfiguration, load_pcd,
                                  pcd_info, write_pcd)
from test_gpu_stream import crc, golden_row
from util import STREAM_CONFIGS, load_stream_frame, stream_gold, stream_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcd_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("pcd")
    for name in stream_names():
        path = d / f"{name}.pcd"
        write_pcd(path, load_stream_frame(name))
        with open(path, "ab") as f:
            f.write(b"\x00" * 3900)  # the reference's files carry trailing bytes after the POINTS records
    return d


def test_pcd_loader_reads_exactly_the_records_into_pinned_memory(pcd_dir):
    for name in stream_names()[::31]:
        want = load_stream_frame(name)
        info = pcd_info(pcd_dir / f"{name}.pcd")
        assert info == dict(n_points=want.shape[0], point_step=16, offsets=(0, 4, 8), n_fields=4)
        got, info2 = load_pcd(pcd_dir / f"{name}.pcd")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        got2, _ = load_pcd(pcd_dir / f"{name}.pcd", pinned=False)
        assert np.array_equal(got2.view(np.uint32), want.view(np.uint32))


def test_pcd_loader_field_layouts_and_errors(tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(777, 5)).astype(np.float32)
    p = tmp_path / "five.pcd"
    write_pcd(p, pts, fields=("intensity", "x", "ring", "y", "z"))
    info = pcd_info(p)
    assert info["point_step"] == 20 and info["offsets"] == (4, 12, 16)
    got, _ = load_pcd(p, pinned=False)
    assert np.array_equal(got, pts)
    (tmp_path / "ascii.pcd").write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\n"
                                        "HEIGHT 1\nPOINTS 1\nDATA ascii\n0 0 0\n")
    with pytest.raises(LpxError):
        pcd_info(tmp_path / "ascii.pcd")
    (tmp_path / "noz.pcd").write_bytes(b"VERSION 0.7\nFIELDS x y\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\n"
                                       b"POINTS 1\nDATA binary\n" + b"\0" * 8)
    with pytest.raises(LpxError):
        pcd_info(tmp_path / "noz.pcd")
    trunc = tmp_path / "trunc.pcd"
    write_pcd(trunc, pts[:, :4])
    with open(trunc, "r+b") as f:
        f.truncate(trunc.stat().st_size - 100)
    with pytest.raises(LpxError):
        load_pcd(trunc)
    with pytest.raises(LpxError):
        pcd_info(tmp_path / "missing.pcd")


@pytest.mark.parametrize("cname", list(STREAM_CONFIGS))
def test_feeder_streams_all_154_files_to_the_goldens(pcd_dir, cname):
    """BASELINE configs[3] end to end: files -> pinned -> H2D -> chains of 32 -> D2H, two buffer sets"""
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    names = stream_names()
    feeder = Feeder([pcd_dir / f"{n}.pcd" for n in names])
    bctx = Context(0, batch=32)
    try:
        assert feeder.n_frames == 154
        assert np.array_equal(feeder.frame(7).view(np.uint32), load_stream_frame(names[7]).view(np.uint32))
        ids = np.arange(154)
        out = feeder.run(bctx, ids, scfg, ccfg)
        out2 = feeder.run(bctx, ids[::-1], scfg, ccfg)  # again, other order: buffers and events are reused
    finally:
        bctx.close()
    g = stream_gold()
    for o, order in ((out, ids), (out2, ids[::-1])):
        for j, fid in enumerate(order):
            ng, no, nc, status = (int(v) for v in o["counts"].array[j])
            assert status == 0
            res = dict(n_ground=ng, n_obstacle=no, n_clusters=nc,
                       labels=o["labels"].array[j, :int(g["n"][fid])], obstacle_idx=o["obstacle_idx"].array[j, :no],
                       cluster_labels=o["cluster_labels"].array[j, :no], planes=o["planes"].array[j].reshape(-1, 4))
            assert golden_row(res) == [int(v) for v in g[cname][fid]], names[fid]
            gi = o["ground_idx"].array[j, :ng]
            assert (o["labels"].array[j, gi] == 1).all()
    feeder.close()


def test_feeder_short_runs_and_single_frame_parity(pcd_dir, ctx):
    names = stream_names()[:5]
    feeder = Feeder([pcd_dir / f"{n}.pcd" for n in names])
    scfg, ccfg = SegmentationConfiguration(), ClusteringConfiguration()
    bctx = Context(0, batch=2)  # chains of 2: 1, 2, 3 chains, a ragged tail
    try:
        for ids in ([3], [0, 1], [4, 2, 0], [1, 1, 1, 1, 1]):
            out = feeder.run(bctx, ids, scfg, ccfg)
            for j, fid in enumerate(ids):
                want = ctx.segment_cluster(load_stream_frame(names[fid]), scfg, ccfg)
                ng, no, nc, status = (int(v) for v in out["counts"].array[j])
                assert (status, nc, ng, no) == (0, want["n_clusters"], len(want["ground_idx"]), len(want["obstacle_idx"]))
                assert np.array_equal(out["labels"].array[j, :len(want["labels"])], want["labels"])
                assert np.array_equal(out["ground_idx"].array[j, :ng], want["ground_idx"])
                assert np.array_equal(out["obstacle_idx"].array[j, :no], want["obstacle_idx"])
                assert np.array_equal(out["cluster_labels"].array[j, :no], want["cluster_labels"])
                assert np.array_equal(out["planes"].array[j].reshape(-1, 4).view(np.uint32), want["planes"].view(np.uint32))
        with pytest.raises(LpxError):
            feeder.run(bctx, [9], scfg, ccfg)
    finally:
        bctx.close()
        feeder.close()


@pytest.mark.parametrize("n_ctx,B", [(3, 8), (4, 16), (7, 32)])
def test_feeder_over_several_contexts_gives_the_single_context_results(pcd_dir, n_ctx, B):
    """lpx_feeder_run_multi: chain k on context k % n_ctx, one pipeline (copy streams, buffer sets, host thread) per
    context; all 154 frames against the goldens, then a ragged second run on the same lanes (more contexts than
    chains included) against the first"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    names = stream_names()
    feeder = Feeder([pcd_dir / f"{n}.pcd" for n in names])
    ctxs = [Context(0, batch=B) for _ in range(n_ctx)]
    try:
        ids = np.arange(154)
        out = feeder.run(ctxs, ids, scfg, ccfg)
        few = np.array([153, 0, 77, 5, 5, 100, 31][: B + 3])
        out2 = feeder.run(ctxs, few, scfg, ccfg)
        with pytest.raises(LpxError):
            feeder.run([ctxs[0], ctxs[0]], ids, scfg, ccfg)
    finally:
        for c in ctxs:
            c.close()
    g = stream_gold()
    for j, fid in enumerate(ids):
        ng, no, nc, status = (int(v) for v in out["counts"].array[j])
        assert status == 0
        res = dict(n_ground=ng, n_obstacle=no, n_clusters=nc,
                   labels=out["labels"].array[j, :int(g["n"][fid])], obstacle_idx=out["obstacle_idx"].array[j, :no],
                   cluster_labels=out["cluster_labels"].array[j, :no], planes=out["planes"].array[j].reshape(-1, 4))
        assert golden_row(res) == [int(v) for v in g[cname][fid]], names[fid]
    for j, fid in enumerate(few):
        assert np.array_equal(out2["counts"].array[j], out["counts"].array[fid])
        ng, no = int(out["counts"].array[fid, 0]), int(out["counts"].array[fid, 1])
        for key, cnt in (("labels", int(g["n"][fid])), ("ground_idx", ng), ("obstacle_idx", no), ("cluster_labels", no)):
            assert np.array_equal(out2[key].array[j, :cnt], out[key].array[fid, :cnt]), (key, fid)
        assert np.array_equal(out2["planes"].array[j].view(np.uint32), out["planes"].array[fid].view(np.uint32))
    feeder.close()


def test_feeder_results_in_ordinary_memory_equal_the_pinned_ones(pcd_dir):
    """result arrays in ordinary (pageable) host memory -- what a caller that never heard of lpx_host_alloc passes --
    receive the same bytes as pinned ones, and nothing is written past the exact sizes: ragged frame list, one and
    several contexts"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    names = stream_names()[:24]
    feeder = Feeder([pcd_dir / f"{n}.pcd" for n in names])
    ctxs = [Context(0, batch=8) for _ in range(3)]
    try:
        ids = np.array([23, 0, 7, 7, 12, 1, 19, 5, 3, 22, 8, 8, 8, 14, 2, 11, 6, 21, 9])
        for who in (ctxs[0], ctxs):
            a = feeder.run(who, ids, scfg, ccfg)
            b = feeder.run(who, ids, scfg, ccfg, pinned=False)
            assert np.array_equal(a["counts"].array, b["counts"].array)
            assert np.array_equal(a["planes"].array.view(np.uint32), b["planes"].array.view(np.uint32))
            for j in range(len(ids)):
                n = feeder.info[int(ids[j])]["n_points"]
                ng, no = a["counts"].array[j, 0], a["counts"].array[j, 1]
                assert np.array_equal(a["labels"].array[j, :n], b["labels"].array[j, :n])
                assert np.array_equal(a["ground_idx"].array[j, :ng], b["ground_idx"].array[j, :ng])
                assert np.array_equal(a["obstacle_idx"].array[j, :no], b["obstacle_idx"].array[j, :no])
                assert np.array_equal(a["cluster_labels"].array[j, :no], b["cluster_labels"].array[j, :no])
                # nothing is written past the exact sizes (the arrays start zeroed)
                assert not b["ground_idx"].array[j, ng:].any() and not a["ground_idx"].array[j, ng:].any()
    finally:
        for c in ctxs:
            c.close()
        feeder.close()


@pytest.mark.parametrize("n_ctx", [1, 2])
def test_feeder_reports_a_flagged_frame_after_finishing_every_other_frame(pcd_dir, tmp_path, n_ctx):
    """one file with a NaN coordinate in the middle of a run: the run returns LPX_ERR_RANGE naming that frame, but only
    after every chain of every lane has been processed and drained -- counts[] holds every frame's status (the header's
    promise) and every other frame's results equal an undisturbed run's"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    names = stream_names()[:12]
    bad = load_stream_frame(names[5]).copy()
    bad[1000, 1] = np.nan
    bad_path = tmp_path / "bad.pcd"
    write_pcd(bad_path, bad)
    paths = [pcd_dir / f"{n}.pcd" for n in names]
    good = Feeder(paths)
    poisoned = Feeder(paths[:5] + [bad_path] + paths[6:])
    ctxs = [Context(0, batch=2) for _ in range(n_ctx)]
    try:
        ids = np.arange(12)  # six chains of two; the flagged frame sits in the third
        who = ctxs[0] if n_ctx == 1 else ctxs
        want = good.run(who, ids, scfg, ccfg)
        out = good.run(who, ids, scfg, ccfg)  # (arrays of the right shape, refilled below)
        for k in ("labels", "ground_idx", "obstacle_idx", "cluster_labels", "planes", "counts"):
            out[k].array[...] = 0
        with pytest.raises(LpxError) as e:
            poisoned.run(who, ids, scfg, ccfg, out)
        assert e.value.code == -2 and "frame 5 " in str(e.value), str(e.value)  # LPX_ERR_RANGE
        status = out["counts"].array[:, 3]
        assert status[5] == 2 and not status[np.arange(12) != 5].any(), status.tolist()
        for j in range(12):
            if j == 5:
                continue
            assert np.array_equal(out["counts"].array[j], want["counts"].array[j]), j
            n, no = good.info[j]["n_points"], int(want["counts"].array[j, 1])
            assert np.array_equal(out["labels"].array[j, :n], want["labels"].array[j, :n]), j
            assert np.array_equal(out["cluster_labels"].array[j, :no], want["cluster_labels"].array[j, :no]), j
    finally:
        for c in ctxs:
            c.close()
        good.close()
        poisoned.close()
