#!/usr/bin/env python3
"""bench.py -- throughput of the segmentation + clustering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (lpx_segment_cluster_batch_device: Segmenter::segment followed by
Clusterer::cluster with the obstacle cloud kept on the device, `--batch` frames per launch chain) over
one batch of frames whose points are already resident in HBM.  Workload = BASELINE.json configs[1]: real 120k-point KITTI frames
(tests/golden/frames.npz, bit-identical to the reference's data/*.pcd), 6 segments, 5 plane-fit
iterations, FEC d = 0.5 m (distance_squared 0.25), quality 0.5.  Frames are independent, so with N
GPUs every rank runs its own batch (frame i -> GPU i mod N, "weak" scaling) and there is no
data-path collective; RCCL is used only for the barrier and the max-over-ranks time.

Prints ONE JSON line (rank 0) with the metric, a `roofline` object for the kernel that fills the device
(HIP-event times measured live: the same K steps under load, and one step with every chain alone on the
device) and, at N = 1, a `cpu_baseline` object: the oracle restatement of the reference path timed on this
host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
# queue serialise.  The bench keeps many frames in flight on separate streams, so ask for 16 queues.
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)

SEG = dict(number_of_planar_partitions=6, number_of_iterations=5)
CLU = dict(distance_squared=0.25, cluster_quality=0.5)


# stage -> kernel that dominates it (names as rocprofv3 prints them), for the PMC traffic lookup
STAGE_KERNEL = {"ingest": "ingest_kernel", "xsort": "radix_scatter_kernel<unsigned int, true>", "gather": "gather_kernel",
                "zsort": "radix_scatter_kernel<unsigned long, false>", "seeds": "seed_kernel",
                "plane_passes": "plane_single_kernel", "compact": "compact_kernel", "kd_build": "kd_block_kernel",
                "cc_hook": "cc_hook_kernel", "neighbours": "nb_group_kernel",
                "components": "radix_scatter_kernel<unsigned int, true>", "replay": "replay_lds_kernel",
                "labels": "relabel_kernel"}


def algorithmic_bytes(stage, N, M, E, I, P, E_replay=None):
    """Algorithmic HBM bytes of `stage` for ONE frame with N points, M obstacle points, E neighbour-list
    entries, E_replay entries in the lists of the points the reference would expand ("Kernels and
    their algorithmic bytes").  A launch group of a batched chain processes frames_per_launch frames."""
    if E_replay is None:
        E_replay = E
    return {
        "ingest": N * (16 + 12 + 8),                 # AoS read, SoA write, (key, index) write
        "xsort": 4 * N * (8 + 8 + 8),                # per pass: histogram read, scatter read + write of 8 B pairs
        "gather": N * (4 + 12 + 12 + 8),             # index, gather, x-sorted SoA, (segment, z) key
        "zsort": 5 * N * (8 + 8 + 8),
        "seeds": N * 4 + P * 64,                     # the selection kernel reads the x-sorted z once
        "plane_passes": N * 12 + N,                  # the SoA is read once and stays in registers; flag write
        "compact": N * (1 + 4 + 4 + 4) + M * (12 + 16),  # flag, index, label, list, obstacle SoA + kd nodes
        "kd_build": M * 16 * 2 * 17,                 # ~log2(M) levels, each reads + writes the node array
        "cc_hook": E * 4 + M * (8 + 8),              # list words read, offsets/lengths, parents
        "neighbours": M * 16 + E * 4 + M * 8,        # nodes read once, one word per neighbour written, off/len
        "components": M * (4 + 4 + 4 + 1 + 4 + 8) + 3 * M * 24,
        "replay": E_replay * 4 + M * (8 + 4 + 4 + 4),  # expanded lists, off/len, seed, queue, valid
        "labels": M * (4 + 4 + 4 + 4),
        "groups": M * (4 + 4 + 4) + 2 * M * 24,
    }[stage]


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's dominant kernel from the committed rocprofv3 --pmc summary of this
    same command (profiles/): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE counts half of a coalesced
    read on gfx950 (guides/MI355X_MICROARCH.md, HBM).  None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_g_pmc_fetch_write_per_kernel.json")
    try:
        d = json.load(open(path))
        k = next(v for name, v in d.items() if name.startswith(STAGE_KERNEL[stage][:40]))
        return int((2 * k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"]) * 1024)
    except Exception:
        return None


def frame_ids_for_rank(rank, world, frames_per_step, n_frames):
    """Frame i of the global stream goes to GPU i mod world: rank r owns i = r, r + world, ...;
    the stream cycles over the n_frames available frames."""
    return [(rank + world * j) % n_frames for j in range(frames_per_step)]


def aggregate(elapsed_s, points_per_step, device, world):
    """MAX of the per-rank time and SUM of the per-rank points (the only cross-rank exchange)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    pts = torch.tensor([float(points_per_step)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
    return float(t.item()), float(pts.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=256, help="frames in one step (per GPU)")
    ap.add_argument("--batch", type=int, default=32, help="frames per launch chain (lpx_segment_cluster_batch_device)")
    ap.add_argument("--contexts", type=int, default=8, help="concurrent lpx contexts (HIP streams) per GPU")
    ap.add_argument("--threads", type=int, default=2, help="host threads that enqueue (ctypes releases the GIL)")
    ap.add_argument("--neighbour-words", type=int, default=256, help="neighbour workspace per point (lpx_reserve)")
    ap.add_argument("--single-pass-words", type=int, default=384,
                    help="extra neighbour workspace per point for single-pass lists (lpx_reserve_single_pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lists", action="store_true", help="round-1 path: materialise every radius list (A/B reference)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
    from util import FRAMES, load_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scfg = SegmentationConfiguration(**SEG)
    ccfg = ClusteringConfiguration(**CLU)
    P = SEG["number_of_planar_partitions"]

    # ---- inputs: resident in HBM before the timed region (32-byte PointXYZI records, one pitched array) ----
    host_frames = [load_frame(f) for f in FRAMES]
    F = args.frames_per_step
    B = max(1, min(args.batch, F))
    my_ids = frame_ids_for_rank(rank, world, F, len(host_frames))
    pitch = max(hf.shape[0] for hf in host_frames)
    host_in = np.zeros((F, pitch, 8), np.float32)
    for j, fid in enumerate(my_ids):
        host_in[j, :host_frames[fid].shape[0], :4] = host_frames[fid]
    d_pts = torch.from_numpy(host_in).to(dev)
    del host_in
    n_points = np.array([host_frames[fid].shape[0] for fid in my_ids], np.uint32)
    chains = [(k, min(k + B, F)) for k in range(0, F, B)]  # frames [lo, hi) of every launch chain
    C = max(1, min(args.contexts, len(chains)))
    ctxs = [Context(local_rank, batch=B) for _ in range(C)]
    for c in ctxs:
        if args.lists:
            c.set_neighbour_mode("lists")
            c.reserve_single_pass(args.single_pass_words)
        c.reserve(pitch, args.neighbour_words)
    d_labels = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    d_gidx = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    d_oidx = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    d_planes = torch.empty((F, 4 * P), dtype=torch.float32, device=dev)
    d_clabels = torch.empty((F, pitch), dtype=torch.int32, device=dev)
    d_counts = torch.zeros((F, 4), dtype=torch.int32, device=dev)
    points_per_step = int(n_points.sum())

    import concurrent.futures
    T = max(1, min(args.threads, C))
    pool = concurrent.futures.ThreadPoolExecutor(T) if T > 1 else None

    def enqueue(tid):
        # thread tid owns contexts tid, tid + T, ... and therefore chains k with (k % C) % T == tid
        torch.cuda.set_device(local_rank)
        for k, (lo, hi) in enumerate(chains):
            if (k % C) % T != tid:
                continue
            ctxs[k % C].segment_cluster_batch_device(n_points[lo:hi], d_pts[lo].data_ptr(), 32, pitch, scfg, ccfg,
                                                     d_labels[lo].data_ptr(), d_gidx[lo].data_ptr(),
                                                     d_oidx[lo].data_ptr(), d_planes[lo].data_ptr(),
                                                     d_clabels[lo].data_ptr(), d_counts[lo].data_ptr())

    def step():
        if pool is None:
            enqueue(0)
        else:
            list(pool.map(enqueue, range(T)))

    def sync():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    counts = d_counts.cpu().numpy().view(np.uint32)
    if (counts[:, 3] != 0).any():
        raise SystemExit(f"device status != 0: {counts[:, 3].tolist()}")

    elapsed, total_points_per_step = aggregate(elapsed, points_per_step, dev, world)

    # ---- stage times: HIP-event pairs around every stage (lpx_profile_*), on the streams the kernels run on ----
    # (1) the same K steps under the same load as the timed region: what a launch group costs while eleven other
    #     chains compete for the device (mostly waiting for free CUs);
    # (2) every chain of one step alone on the device: the launch duration of the kernels themselves.
    # The roofline object is for the kernel that dominates (2) -- the one that fills the device -- and also
    # carries its average duration under load.
    roofline = None
    stage_ms = {}
    if rank == 0:
        def profiled(run):
            for c in ctxs:
                c.profile_enable(True)
            run()
            sync()
            ms_tot, n_tot = {}, {}
            for c in ctxs:
                for k, (ms, cnt) in c.profile_read().items():
                    ms_tot[k] = ms_tot.get(k, 0.0) + ms
                    n_tot[k] = n_tot.get(k, 0) + cnt
                c.profile_enable(False)
            return ms_tot, n_tot

        def loaded():
            for _ in range(args.steps):
                step()

        def isolated():
            for k, (lo, hi) in enumerate(chains):
                c = ctxs[k % C]
                c.segment_cluster_batch_device(n_points[lo:hi], d_pts[lo].data_ptr(), 32, pitch, scfg, ccfg,
                                               d_labels[lo].data_ptr(), d_gidx[lo].data_ptr(), d_oidx[lo].data_ptr(),
                                               d_planes[lo].data_ptr(), d_clabels[lo].data_ptr(),
                                               d_counts[lo].data_ptr())
                c.synchronize()

        stage_ms, launches = profiled(loaded)
        iso_ms, iso_launches = profiled(isolated)
        dom = max(iso_ms, key=iso_ms.get)
        avg_ms = iso_ms[dom] / max(1, iso_launches[dom])
        avg_ms_loaded = stage_ms[dom] / max(1, launches[dom])
        # one launch (group) of a stage covers the B frames of a chain: frame-averaged sizes of this rank's
        # batch times the frames per chain; list sizes come from the device counters of the frame slots
        frames_per_launch = float(np.mean([hi - lo for lo, hi in chains]))
        Nn = float(n_points.mean())
        Mm = float(counts[:, 1].mean())
        fst = [c.frame_stats(slot) for c in ctxs for slot in range(B)]
        E = float(np.mean([f["neighbour_entries"] for f in fst]))
        E_replay = float(np.mean([f["replay_entries"] for f in fst]))
        algo = frames_per_launch * algorithmic_bytes(dom, Nn, Mm, E, SEG["number_of_iterations"], P, E_replay)
        achieved = algo / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "stage": dom, "achieved": round(achieved, 3),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": pmc_traffic(dom), "avg_launch_ms": round(avg_ms, 5),
                    "avg_launch_ms_under_load": round(avg_ms_loaded, 5), "algorithmic_bytes_per_launch": int(algo),
                    "frames_per_launch": frames_per_launch,
                    "stage_ms_per_launch_alone": {k: round(v / max(1, iso_launches[k]), 5) for k, v in iso_ms.items()}}

    # ---- CPU baseline: the oracle restatement on this host, bounded sample (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        oscfg = oracle.SegCfg(**SEG)
        occfg = oracle.CluCfg(CLU["distance_squared"], CLU["cluster_quality"])
        done_pts, t_cpu, passes = 0, 0.0, 0
        while t_cpu < 10.0:
            hf = host_frames[passes % len(host_frames)]
            a = time.perf_counter()
            r = oracle.segment(hf, oscfg)
            oracle.cluster(hf[r["obstacle_idx"]], occfg)
            t_cpu += time.perf_counter() - a
            done_pts += hf.shape[0]
            passes += 1
        cpu = {"value": round(done_pts / t_cpu / 1e6, 4), "unit": "Mpts/s", "cores": 1, "kind": "port",
               "sample": f"{passes} frame passes of the same 3 KITTI frames (segment+cluster, oracle/lidar_oracle.c, "
                         f"{t_cpu:.1f} s)", "host_cpus": os.cpu_count()}

    if rank == 0:
        value = total_points_per_step * args.steps / elapsed / 1e6
        line = {
            "metric": "Mpts/s seg+cluster (120k-pt frame)",
            "value": round(value, 3),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real KITTI frames (committed fixture of the reference's data/*.pcd), random-free",
            "config": {"workload": "configs[1]: 120k-pt KITTI frames, 6 segments, 5 iters, FEC d=0.5 m q=0.5",
                       "frames_per_step_per_gpu": F, "frames_per_launch_chain": B, "contexts_per_gpu": C,
                       "host_threads_per_gpu": T,
                       "hip_hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "points_per_step": int(total_points_per_step),
                       "frames_per_s": round(F * world * args.steps / elapsed, 2)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stage_ms_per_frame_under_load": {k: round(v / (args.steps * F), 5) for k, v in stage_ms.items()},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
