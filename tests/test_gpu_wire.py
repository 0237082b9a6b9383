"""N4: sensor_msgs/PointCloud2 wire format on the device, through the C-ABI.

Ingest: the message's data[] buffer with its point_step and x / y / z field offsets goes to the GPU as it is
(the reference decodes it on the host first, src/conversions.cpp:62-85); results must equal the PCL-record path
and the oracle.  Egress: the two recoloured PointXYZRGBL clouds of src/processor.cpp:152-163, byte for byte
against the numpy restatement oracle.coloured_records."""
import numpy as np
import pytest

import oracle
from lidar_processing_amd import ClusteringConfiguration, Context, LpxError, SegmentationConfiguration
from util import FRAMES, load_frame

pytestmark = pytest.mark.gpu

SEG = dict(number_of_planar_partitions=6, number_of_iterations=5)
CLU = dict(distance_squared=0.25, cluster_quality=0.5)


def message(pts, point_step, offsets, seed=0):
    """a PointCloud2-style byte buffer: float32 x, y, z at `offsets` of every point_step-byte record, the rest of
    the record filled with other fields' bytes (noise)"""
    n = pts.shape[0]
    rng = np.random.default_rng(seed)
    buf = rng.integers(0, 256, (n, point_step), dtype=np.uint8)
    for k, o in enumerate(offsets):
        buf[:, o:o + 4] = np.ascontiguousarray(pts[:, k]).view(np.uint8).reshape(n, 4)
    return buf


@pytest.mark.parametrize("point_step,offsets", [(32, (0, 4, 8)),      # the reference's dataloader: sizeof(PointXYZI)
                                                (16, (0, 4, 8)),      # x y z intensity, packed
                                                (20, (4, 8, 12)),     # a leading field
                                                (48, (8, 0, 40)),     # fields in another order, Ouster-sized records
                                                (22, (0, 4, 8)),      # point_step not a multiple of 4
                                                (19, (1, 6, 13)),     # nothing aligned
                                                (12, (0, 4, 8))])
def test_pointcloud2_buffer_ingested_directly(ctx, point_step, offsets):
    pts = load_frame(FRAMES[1])[:70_001]
    buf = message(pts, point_step, offsets, seed=point_step)
    scfg, ccfg = SegmentationConfiguration(**SEG), ClusteringConfiguration(**CLU)
    out = ctx.segment_cluster_fields(buf, point_step, offsets, pts.shape[0], scfg, ccfg)
    ref = ctx.segment_cluster(pts, scfg, ccfg)
    for k in ("labels", "ground_idx", "obstacle_idx", "cluster_labels"):
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(out["planes"].view(np.uint32), ref["planes"].view(np.uint32))
    assert out["n_clusters"] == ref["n_clusters"]
    r = oracle.segment(pts, oracle.SegCfg(**SEG))
    assert np.array_equal(out["labels"], r["labels"]) and np.array_equal(out["obstacle_idx"], r["obstacle_idx"])
    want, wn = oracle.cluster(pts[r["obstacle_idx"]], oracle.CluCfg(**CLU))
    assert wn == out["n_clusters"] and np.array_equal(out["cluster_labels"], want)
    seg_only = ctx.segment_cluster_fields(buf, point_step, offsets, pts.shape[0], scfg)
    assert np.array_equal(seg_only["labels"], r["labels"]) and np.array_equal(seg_only["ground_idx"], r["ground_idx"])


def test_pointcloud2_argument_errors(ctx):
    pts = load_frame(FRAMES[0])[:1000]
    buf = message(pts, 16, (0, 4, 8))
    scfg = SegmentationConfiguration()
    with pytest.raises(LpxError):
        ctx.segment_cluster_fields(buf, 16, (0, 4, 13), 1000, scfg)  # z would read past the record
    with pytest.raises(LpxError):
        ctx.segment_cluster_fields(buf, 3, (0, 0, 0), 1000, scfg)
    out = ctx.segment_cluster_fields(buf, 16, (0, 4, 8), 0, scfg)  # empty message
    assert out["labels"].shape == (0,)
    bad = buf.copy()
    bad[7, 4:8] = np.array([np.nan], np.float32).view(np.uint8)
    with pytest.raises(LpxError) as e:
        ctx.segment_cluster_fields(bad, 16, (0, 4, 8), 1000, scfg)
    assert e.value.code == -2


@pytest.mark.parametrize("frame", FRAMES)
def test_coloured_clouds_match_the_recolour_copy(ctx, frame):
    pts = load_frame(frame)
    scfg = SegmentationConfiguration(**SEG)
    labels, gi, oi, _ = ctx.segment(pts, scfg)
    g, o = ctx.coloured_clouds(len(gi), len(oi))
    r = oracle.segment(pts, oracle.SegCfg(**SEG))
    assert np.array_equal(g, oracle.coloured_records(pts, r["ground_idx"], True))
    assert np.array_equal(o, oracle.coloured_records(pts, r["obstacle_idx"], False))
    # also after the fused call, and after the regrouping (which must not clobber the resident index lists)
    out = ctx.segment_cluster(pts, scfg, ClusteringConfiguration(**CLU))
    ctx.cluster_groups(len(out["obstacle_idx"]), out["n_clusters"])
    g2, o2 = ctx.coloured_clouds(len(out["ground_idx"]), len(out["obstacle_idx"]))
    assert np.array_equal(g2, g) and np.array_equal(o2, o)


def test_coloured_clouds_edge_cases(ctx):
    empty = np.zeros((0, 4), np.float32)
    ctx.segment(empty, SegmentationConfiguration())
    g, o = ctx.coloured_clouds(0, 0)
    assert g.shape == (0, 32) and o.shape == (0, 32)
    rng = np.random.default_rng(5)
    flat = np.zeros((500, 4), np.float32)  # no seeds -> everything obstacle (Q4)
    flat[:, 0] = rng.random(500) * 30
    flat[:, 2] = -1.7 + rng.random(500) * 0.1
    labels, gi, oi, _ = ctx.segment(flat, SegmentationConfiguration(number_of_planar_partitions=1))
    assert len(gi) == 0 and len(oi) == 500
    g, o = ctx.coloured_clouds(0, 500)
    assert np.array_equal(o, oracle.coloured_records(flat, oi, False))


def test_coloured_clouds_refuse_a_workspace_that_was_reallocated():
    """lpx_reserve (like every growth of the workspace) frees the arena the segmented cloud lived in: the records of the
    earlier lpx_segment are gone, and lpx_coloured_clouds must say so instead of colouring the zeroed new arena"""
    from lidar_processing_amd import LpxError
    pts = load_frame(FRAMES[0])[:20_000]
    c = Context(0)
    try:
        c.reserve(20_000)
        _, gi, oi, _ = c.segment(pts, SegmentationConfiguration(**SEG))
        g, o = c.coloured_clouds(len(gi), len(oi))  # fine: the cloud is resident
        assert np.array_equal(g, oracle.coloured_records(pts, gi, True))
        c.reserve(60_000)  # reallocates
        with pytest.raises(LpxError) as e:
            c.coloured_clouds(len(gi), len(oi))
        assert e.value.code == -1  # LPX_ERR_ARG
        _, gi, oi, _ = c.segment(pts, SegmentationConfiguration(**SEG))
        c.reserve(30_000)  # no growth: nothing moves, the cloud stays valid
        g, o = c.coloured_clouds(len(gi), len(oi))
        assert np.array_equal(o, oracle.coloured_records(pts, oi, False))
    finally:
        c.close()


def test_coloured_clouds_batch_device(ctx):
    import torch
    dev = torch.device("cuda:0")
    clouds = [load_frame(f)[:40_000 + 1000 * i] for i, f in enumerate(FRAMES)]
    B, pitch, P = len(clouds), 43_000, SEG["number_of_planar_partitions"]
    host = np.zeros((B, pitch, 4), np.float32)
    for b, c in enumerate(clouds):
        host[b, :c.shape[0]] = c
    d_pts = torch.from_numpy(host).to(dev)
    mk = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)  # noqa: E731
    d_labels, d_g, d_o, d_cl, d_cnt = mk(B, pitch), mk(B, pitch), mk(B, pitch), mk(B, pitch), mk(B, 4)
    d_planes = mk(B, 4 * P, dt=torch.float32)
    d_grec, d_orec = mk(B, pitch, 32, dt=torch.uint8), mk(B, pitch, 32, dt=torch.uint8)
    torch.cuda.synchronize()
    bctx = Context(0, batch=B)
    try:
        n = np.array([c.shape[0] for c in clouds], np.uint32)
        bctx.segment_cluster_batch_device(n, d_pts.data_ptr(), 16, pitch, SegmentationConfiguration(**SEG),
                                          ClusteringConfiguration(**CLU), d_labels.data_ptr(), d_g.data_ptr(),
                                          d_o.data_ptr(), d_planes.data_ptr(), d_cl.data_ptr(), d_cnt.data_ptr())
        bctx.check(bctx._L.lpx_coloured_clouds_batch_device(bctx._h, B, pitch, d_g.data_ptr(), d_o.data_ptr(),
                                                            d_grec.data_ptr(), d_orec.data_ptr()))
        bctx.synchronize()
    finally:
        bctx.close()
    cnt = d_cnt.cpu().numpy()
    for b, c in enumerate(clouds):
        r = oracle.segment(c, oracle.SegCfg(**SEG))
        ng, no = int(cnt[b, 0]), int(cnt[b, 1])
        assert (ng, no) == (len(r["ground_idx"]), len(r["obstacle_idx"]))
        assert np.array_equal(d_grec[b, :ng].cpu().numpy(), oracle.coloured_records(c, r["ground_idx"], True))
        assert np.array_equal(d_orec[b, :no].cpu().numpy(), oracle.coloured_records(c, r["obstacle_idx"], False))


@pytest.mark.parametrize("B", [1, 3])
def test_record_copy_lets_the_caller_recycle_its_input(B):
    """lpx_set_record_copy: by default lpx_coloured_clouds*_device read x, y, z from the INPUT array of
    the segmentation call (include/lpx.h, LIFETIME OF THE INPUT), so a caller that overwrites its upload buffer first gets
    the new bytes; with the copy on, the segmentation keeps the coordinates in its arena and the records are those of
    the segmented cloud -- same labels, lists and clusters either way."""
    import torch
    dev = torch.device("cuda:0")
    clouds = [load_frame(f)[:30_000 + 500 * i] for i, f in enumerate(FRAMES[:B])]
    pitch, P = 32_000, SEG["number_of_planar_partitions"]
    host = np.zeros((B, pitch, 4), np.float32)
    for b, c in enumerate(clouds):
        host[b, :c.shape[0]] = c
    n = np.array([c.shape[0] for c in clouds], np.uint32)
    want = [oracle.segment(c, oracle.SegCfg(**SEG)) for c in clouds]
    for keep in (True, False):
        d_pts = torch.from_numpy(host).to(dev)
        mk = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)  # noqa: E731
        d_labels, d_g, d_o, d_cl, d_cnt = mk(B, pitch), mk(B, pitch), mk(B, pitch), mk(B, pitch), mk(B, 4)
        d_planes = mk(B, 4 * P, dt=torch.float32)
        d_grec, d_orec = mk(B, pitch, 32, dt=torch.uint8), mk(B, pitch, 32, dt=torch.uint8)
        torch.cuda.synchronize()
        c = Context(0, batch=B)
        try:
            c.set_record_copy(keep)
            scfg, ccfg = SegmentationConfiguration(**SEG), ClusteringConfiguration(**CLU)
            if B == 1:
                c.segment_cluster_device(d_pts.data_ptr(), 16, int(n[0]), scfg, ccfg, d_labels.data_ptr(), d_g.data_ptr(),
                                         d_o.data_ptr(), d_planes.data_ptr(), d_cl.data_ptr(), d_cnt.data_ptr())
            else:
                c.segment_cluster_batch_device(n, d_pts.data_ptr(), 16, pitch, scfg, ccfg, d_labels.data_ptr(),
                                               d_g.data_ptr(), d_o.data_ptr(), d_planes.data_ptr(), d_cl.data_ptr(),
                                               d_cnt.data_ptr())
            c.synchronize()
            d_pts.fill_(7.0)  # the caller recycles its upload buffer
            torch.cuda.synchronize()
            if B == 1:
                c.check(c._L.lpx_coloured_clouds_device(c._h, d_g.data_ptr(), d_o.data_ptr(), d_grec.data_ptr(),
                                                        d_orec.data_ptr()))
            else:
                c.check(c._L.lpx_coloured_clouds_batch_device(c._h, B, pitch, d_g.data_ptr(), d_o.data_ptr(),
                                                              d_grec.data_ptr(), d_orec.data_ptr()))
            c.synchronize()
        finally:
            c.close()
        cnt = d_cnt.cpu().numpy()
        for b, cl in enumerate(clouds):
            r = want[b]
            ng, no = int(cnt[b, 0]), int(cnt[b, 1])
            assert (ng, no) == (len(r["ground_idx"]), len(r["obstacle_idx"]))
            assert np.array_equal(d_labels[b, :cl.shape[0]].cpu().numpy().astype(np.uint32), r["labels"])
            got = d_orec[b, :no].cpu().numpy()
            if keep:
                assert np.array_equal(d_grec[b, :ng].cpu().numpy(), oracle.coloured_records(cl, r["ground_idx"], True))
                assert np.array_equal(got, oracle.coloured_records(cl, r["obstacle_idx"], False))
            else:  # the documented default: the records are read where the caller left them -- now all sevens
                assert np.array_equal(got[:, :12].view(np.float32), np.full((no, 3), 7.0, np.float32))
