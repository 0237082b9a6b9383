"""GPU half of the 154-frame stream (BASELINE.json configs[3]) and the bench-shaped concurrency test.

* every frame of the reference's data/*.pcd through the multi-frame launch chain, both configurations of
  tests/golden/stream_golden.npz: counts and CRC-32 of the segmentation labels, the obstacle order, the plane
  words and the cluster labels (the latter produced by the REFERENCE's own kd-tree build);
* the shape bench.py runs (8 batch contexts x 32 frame slots, 2 enqueue threads, 3 rounds, 256 frames in
  flight): every frame's five outputs equal the single-frame path and the goldens."""
import concurrent.futures
import zlib

import numpy as np
import pytest

from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import STREAM_CONFIGS, load_stream_frame, stream_gold, stream_names

pytestmark = pytest.mark.gpu


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def stream():
    return [load_stream_frame(n) for n in stream_names()]


class Pitched:
    """F frames resident in HBM as pitched arrays, like bench.py keeps them"""

    def __init__(self, frames, P):
        import torch
        self.torch = torch
        dev = torch.device("cuda:0")
        self.F = len(frames)
        self.n = np.array([f.shape[0] for f in frames], np.uint32)
        self.pitch = int(self.n.max())
        host = np.zeros((self.F, self.pitch, 8), np.float32)  # 32-byte PointXYZI records
        for j, f in enumerate(frames):
            host[j, :f.shape[0], :4] = f
        self.pts = torch.from_numpy(host).to(dev)
        self.P = P
        self.labels = torch.zeros((self.F, self.pitch), dtype=torch.int32, device=dev)
        self.gidx = torch.zeros((self.F, self.pitch), dtype=torch.int32, device=dev)
        self.oidx = torch.zeros((self.F, self.pitch), dtype=torch.int32, device=dev)
        self.planes = torch.zeros((self.F, 4 * P), dtype=torch.float32, device=dev)
        self.clab = torch.zeros((self.F, self.pitch), dtype=torch.int32, device=dev)
        self.counts = torch.zeros((self.F, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

    def clear(self):
        for t in (self.labels, self.gidx, self.oidx, self.clab, self.counts):
            t.fill_(-3)
        self.planes.fill_(9.0)
        self.torch.cuda.synchronize()

    def enqueue(self, ctx, lo, hi, scfg, ccfg):
        ctx.segment_cluster_batch_device(self.n[lo:hi], self.pts[lo].data_ptr(), 32, self.pitch, scfg, ccfg,
                                         self.labels[lo].data_ptr(), self.gidx[lo].data_ptr(),
                                         self.oidx[lo].data_ptr(), self.planes[lo].data_ptr(),
                                         self.clab[lo].data_ptr(), self.counts[lo].data_ptr())

    def frame(self, j):
        ng, no, nc, status = (int(v) for v in self.counts[j].cpu().numpy().view(np.uint32))
        n = int(self.n[j])
        return dict(status=status, n_ground=ng, n_obstacle=no, n_clusters=nc,
                    labels=self.labels[j, :n].cpu().numpy().view(np.uint32),
                    ground_idx=self.gidx[j, :ng].cpu().numpy().view(np.uint32),
                    obstacle_idx=self.oidx[j, :no].cpu().numpy().view(np.uint32),
                    planes=self.planes[j].cpu().numpy().reshape(self.P, 4),
                    cluster_labels=self.clab[j, :no].cpu().numpy())


def golden_row(res):
    return [res["n_ground"], res["n_obstacle"], res["n_clusters"], crc(res["labels"].astype(np.uint8)),
            crc(res["obstacle_idx"]), crc(res["cluster_labels"]), crc(res["planes"])]


@pytest.mark.parametrize("cname", list(STREAM_CONFIGS))
def test_stream_154_frames_match_reference_goldens(stream, cname):
    """BASELINE configs[3]: all 154 frames in filename order, 32 per launch chain"""
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    g = stream_gold()
    buf = Pitched(stream, skw["number_of_planar_partitions"])
    buf.clear()
    bctx = Context(0, batch=32)
    try:
        bctx.reserve(buf.pitch)
        for lo in range(0, buf.F, 32):
            buf.enqueue(bctx, lo, min(lo + 32, buf.F), scfg, ccfg)
        bctx.synchronize()
    finally:
        bctx.close()
    bad = []
    for j in range(buf.F):
        res = buf.frame(j)
        if res["status"] != 0 or golden_row(res) != [int(v) for v in g[cname][j]]:
            bad.append((stream_names()[j], res["status"], golden_row(res), g[cname][j].tolist()))
    assert not bad, bad[:3]


def test_list_mode_chains_of_a_fresh_context_refuse_no_frame(stream):
    """the 154 frames in chains of 32 through a FRESH batch context in LIST mode, the order a closed loop of two
    contexts walks them: a chain is a device call, which nothing repeats, so its list workspace must carry the
    reference's own frames from the first call on (round 6's small start refused 17 of them with LPX_ERR_CAPACITY
    while the workspace grew; contexts with more than one frame slot keep the reserve of rounds 1-5) -- every frame's
    status word and outputs against the goldens of the reference build"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    g = stream_gold()
    buf = Pitched(stream, skw["number_of_planar_partitions"])
    bad, chains = [], 0
    for first in (0, 32):
        buf.clear()
        bctx = Context(0, batch=32)
        try:
            bctx.set_neighbour_mode("lists")
            bctx.reserve(buf.pitch)
            k, seen = first, []
            for chain in range(8):
                lo = k % (buf.F - 32 + 1)
                buf.enqueue(bctx, lo, lo + 32, scfg, ccfg)
                bctx.synchronize()
                for j in range(lo, lo + 32):
                    res = buf.frame(j)
                    if res["status"] != 0 or golden_row(res) != [int(v) for v in g[cname][j]]:
                        bad.append((chain, stream_names()[j], res["status"]))
                k += 64
                chains += 1
            grown = bctx.workspace_bytes()[1]
        finally:
            bctx.close()
        assert grown < 16 << 30, grown  # (32 x 379 MB as in rounds 1-5, not a runaway growth)
    assert not bad and chains == 16, bad[:5]


@pytest.mark.parametrize("mode", ["lists", "search"])
def test_stream_through_the_two_calls_of_the_node(stream, mode):
    """the unchanged node's form on all 154 frames in order: segment(), then cluster() on the obstacle cloud it returned,
    one context -- from the second frame on every cluster() finds its clustering enqueued by segment()
    (lpx_set_lookahead); every frame's outputs against the goldens of the reference build"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    g = stream_gold()
    c = Context(0)
    bad = []
    try:
        c.set_neighbour_mode(mode)
        for j, pts in enumerate(stream):
            labels, gi, oi, planes = c.segment(pts, scfg)
            cl, nc = c.cluster(np.ascontiguousarray(pts[oi]), ccfg)
            row = golden_row(dict(n_ground=len(gi), n_obstacle=len(oi), n_clusters=nc, labels=labels, obstacle_idx=oi,
                                  cluster_labels=cl, planes=planes))
            if row != [int(v) for v in g[cname][j]]:
                bad.append((stream_names()[j], row, g[cname][j].tolist()))
        assert c.lookahead_hits() == len(stream) - 1
    finally:
        c.close()
    assert not bad, bad[:3]


@pytest.mark.parametrize("shape", [(256, 32, 8, 2, False), (640, 64, 10, 4, False), (512, 64, 4, 2, True),
                                   (1280, 64, 20, 4, False)],
                         ids=["8x32", "10x64", "4x64-overlap", "20x64"])
def test_bench_shape_contexts_slots_threads(stream, shape):
    """the configurations bench.py times -- round 2's 32-frame chains on 8 contexts, round 3's 64-frame chains on many
    contexts, and lpx_set_overlap (two slot sets per context, the tail of a chain beside the next chain) -- several host
    threads enqueueing, three rounds back to back without a synchronisation in between; afterwards every frame's labels,
    index lists, planes, cluster labels and counts equal the single-frame entry point's"""
    cname = "p6i5_d025q05"
    skw, ckw = STREAM_CONFIGS[cname]
    scfg, ccfg = SegmentationConfiguration(**skw), ClusteringConfiguration(**ckw)
    F, B, C, T, overlap = shape
    ROUNDS = 3
    ids = [(5 * j) % len(stream) for j in range(F)]  # 154 distinct frames spread over the slots, some twice
    buf = Pitched([stream[i] for i in ids], skw["number_of_planar_partitions"])
    chains = [(k, min(k + B, F)) for k in range(0, F, B)]
    ctxs = [Context(0, batch=B) for _ in range(C)]
    try:
        for c in ctxs:
            c.reserve(buf.pitch)
            if overlap:
                c.set_overlap(True)

        def enqueue(tid):
            buf.torch.cuda.set_device(0)
            for k, (lo, hi) in enumerate(chains):
                if (k % C) % T == tid:
                    buf.enqueue(ctxs[k % C], lo, hi, scfg, ccfg)

        with concurrent.futures.ThreadPoolExecutor(T) as pool:
            for r in range(ROUNDS):
                if r == ROUNDS - 1:
                    for c in ctxs:
                        c.synchronize()
                    buf.clear()  # the last round has to produce everything again
                list(pool.map(enqueue, range(T)))
        for c in ctxs:
            c.synchronize()
    finally:
        for c in ctxs:
            c.close()
    g = stream_gold()
    one = Context(0)
    try:
        one.reserve(buf.pitch)
        ref = {}
        for j in range(F):
            res = buf.frame(j)
            assert res["status"] == 0
            assert golden_row(res) == [int(v) for v in g[cname][ids[j]]], stream_names()[ids[j]]
            if ids[j] not in ref:
                ref[ids[j]] = one.segment_cluster(stream[ids[j]], scfg, ccfg)
            want = ref[ids[j]]
            for k in ("labels", "ground_idx", "obstacle_idx", "cluster_labels"):
                assert np.array_equal(res[k], want[k]), (j, k)
            assert np.array_equal(res["planes"].view(np.uint32), want["planes"].view(np.uint32))
            assert res["n_clusters"] == want["n_clusters"]
    finally:
        one.close()


def test_bench_line_verifies_its_own_timed_region():
    """bench.py compares every frame of its timed region's last step (the headline shape: closed loops on every context)
    with the committed goldens and says so in the line; a mismatch makes it exit 1.  Also: the line echoes the
    environment it ran in and the library it loaded (the release build: no LPX_* knobs)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LPX_")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-latency", "--no-inflight", "--no-sub"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    v = line["verified"]
    assert v["frames"] == line["config"]["frames_per_step_per_gpu"] and v["mismatches"] == 0, v
    assert line["completion"]["p99_frame_completion_ms"] > 0 and line["completion"]["frames_in_flight"] > 0
    assert "release build" in line["config"]["env"]["library_build"], line["config"]["env"]
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["verified_mismatches"] == 0
    # the N = 1 line has run the RCCL leg once (world 1, child process) and says which RCCL
    assert "self-test ok" in line["config"]["distributed_backend"], line["config"]["distributed_backend"]
    assert line["config"]["distributed_selftest"]["ok"] is True


@pytest.mark.parametrize("launcher_env", [False, True])
def test_rccl_world_1_selftest(launcher_env):
    """the reporting collectives of the N > 1 line -- barrier, float64 MAX / SUM all-reduce, all-gather on
    device tensors -- executed on RCCL (torch.distributed backend "nccl") in a world of one rank bound to cuda:0,
    through the very functions the N > 1 line uses (bench.aggregate, bench.gather_per_rank).  launcher_env: with the
    environment `python -m torch.distributed.run --nproc-per-node 1` leaves its worker (the agent-store flag and the
    launcher's own port): the self-test must host its own store instead of waiting for the launcher's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    if launcher_env:
        env.update(TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RESTART_COUNT="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29517", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-selftest"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["dist_selftest"]
    assert rec["ok"] is True and rec["world"] == 1 and rec["backend"].startswith("nccl")
    assert rec["rccl_version"].count(".") >= 1 and "self-test ok" in rec["summary"]
    assert rec["gpu_pci"] is None or rec["gpu_pci"].count(":") == 2


def test_two_ranks_on_one_gpu_run_real_frames():
    """SURVEY 8(e), the N > 1 path on REAL frames before an 8-GPU node sees it: `bench.py --gpus 2 --backend gloo
    --device-map 0,0` starts its two ranks itself (before anything touches the GPU), both on GPU 0 (RCCL refuses two ranks
    on one device, hence gloo: the collectives carry a few float64 words of the report either way).  Asserted: rank r got
    the frames r, r + 2, ... of the stream; every frame of every rank's last step equals tests/golden; per_rank has two
    entries; value = sum of the ranks' points / the slowest rank's time.  No scaling claim follows from two ranks on one
    device, and the line says so."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    steps = 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-map",
                        "0,0", "--steps", str(steps), "--warmup", "1", "--workload", "stream", "--frames-per-step", "24",
                        "--batch", "8", "--contexts", "3", "--no-cpu-baseline", "--no-latency", "--no-inflight", "--no-sub"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["distributed_world_size"] == 2 and cfg["ranks_share_gpus"] is True
    assert cfg["device_map"] == [0, 0] and "NOT a scaling measurement" in cfg["distributed_backend"]
    pr = line["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and all(p["gpu_index"] == 0 for p in pr)
    assert pr[0]["first_frame_ids"] == [0, 2, 4, 6] and pr[1]["first_frame_ids"] == [1, 3, 5, 7]  # frame i -> rank i mod 2
    assert all(p["verified_mismatches"] == 0 and p["verified_frames"] == 24 for p in pr), pr
    assert line["verified_mismatches"] == 0 and line["verified_frames"] == 48
    assert cfg["frames_per_step"] == 48 and cfg["points_per_step"] == sum(p["points_per_step"] for p in pr)
    slowest_ms = max(p["ms_per_step"] for p in pr)
    assert abs(line["ms_per_step"] - slowest_ms) <= 1e-3 * slowest_ms + 1e-3
    want = cfg["points_per_step"] / (line["ms_per_step"] * 1e-3) / 1e6
    assert abs(line["value"] - want) <= 2e-3 * want, (line["value"], want)
