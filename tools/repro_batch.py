"""Repeats one batch scenario (optionally next to a noisy neighbour context) and reports frames whose cluster
labels differ from the oracle's (debug aid).  usage: repro_batch.py [reps] [noise 0/1] [modes]"""
import sys, os, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle
from lidar_processing_amd import Context, SegmentationConfiguration, ClusteringConfiguration
from util import synthetic_scene, FRAMES, load_frame
from test_gpu_batch import run_batch, CLU

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
noise = len(sys.argv) > 2 and sys.argv[2] == "1"
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["search", "lists"]
clouds = [synthetic_scene(120_000, 60, 500, seed=7), synthetic_scene(90_000, 40, 400, seed=8)]
seg_kw = dict(number_of_planar_partitions=2, number_of_iterations=3)
want = []
for c in clouds:
    o = oracle.segment(c, oracle.SegCfg(**seg_kw))
    want.append(oracle.cluster(c[o["obstacle_idx"]], oracle.CluCfg(**CLU))[0])
stop = False


def neighbour():
    nctx = Context(0)
    f = load_frame(FRAMES[0])
    s, c = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5), ClusteringConfiguration(0.25, 0.5)
    while not stop:
        nctx.segment_cluster(f, s, c)
    nctx.close()


th = threading.Thread(target=neighbour) if noise else None
if th:
    th.start()
try:
    for mode in modes:
        bad = 0
        bctx = Context(0, batch=2)
        bctx.set_neighbour_mode(mode)
        for rep in range(reps):
            if rep % 50 == 49:  # fresh arenas now and then
                bctx.close()
                bctx = Context(0, batch=2)
                bctx.set_neighbour_mode(mode)
            res = run_batch(bctx, clouds, seg_kw, CLU)
            for b, (r, w) in enumerate(zip(res, want)):
                if not np.array_equal(r["cluster_labels"], w):
                    bad += 1
                    d = np.flatnonzero(r["cluster_labels"] != w)
                    print(mode, "rep", rep, "frame", b, "differs at", d.size, "points; first", d[:8], "got",
                          r["cluster_labels"][d[:8]], "want", w[d[:8]], "n_clusters", r["n_clusters"], int(w.max()) + 1,
                          bctx.frame_stats(b), flush=True)
        bctx.close()
        print(mode, "mismatching frames:", bad, "of", reps * len(clouds), flush=True)
finally:
    stop = True
    if th:
        th.join()

# the host entry point on a long-lived and on fresh single-frame contexts (list path with its capacity retry)
scfg, ccfg = SegmentationConfiguration(**seg_kw), ClusteringConfiguration(**CLU)
kitti = load_frame(FRAMES[1])
long_ctx = Context(0)
bad = 0
for rep in range(reps):
    ctxs = [long_ctx, Context(0)] if rep % 10 == 0 else [long_ctx]
    for cx in ctxs:
        if rep % 3 == 0:
            cx.segment_cluster(kitti[: 20_000 + 997 * (rep % 50)], SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5), ccfg)  # other sizes in between
        for b, (c, w) in enumerate(zip(clouds, want)):
            r = cx.segment_cluster(c, scfg, ccfg)
            if not np.array_equal(r["cluster_labels"], w):
                bad += 1
                d = np.flatnonzero(r["cluster_labels"] != w)
                print("single rep", rep, "fresh" if cx is not long_ctx else "long", "frame", b, "differs at", d.size,
                      "first", d[:8], r["cluster_labels"][d[:8]], w[d[:8]], cx.frame_stats(0), flush=True)
    if len(ctxs) > 1:
        ctxs[1].close()
print("single-frame host path mismatches:", bad, flush=True)
