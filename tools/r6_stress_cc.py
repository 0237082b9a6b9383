import os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from lidar_processing_amd import Context
from util import FRAMES, load_frame
big = np.concatenate([load_frame(f) for f in FRAMES])[:200_000]
r = oracle.segment(big, oracle.SegCfg(number_of_planar_partitions=2, number_of_iterations=3))
obs = np.ascontiguousarray(big[r["obstacle_idx"]][:, :3])
print("obstacles", obs.shape)
for mode in ("lists", "search"):
    c = Context(0); c.set_neighbour_mode(mode); c.reserve(obs.shape[0], 600)
    first = None; bad = 0
    for rep in range(int(sys.argv[1])):
        roots = c.dbg_components(obs, 0.25)
        if first is None: first = roots.copy()
        elif not np.array_equal(first, roots): bad += 1
    print(mode, "reps", sys.argv[1], "differing results", bad, "components", len(np.unique(first)))
    c.close()
# ... and the whole clustering of the same cloud (single-frame context, both modes)
from lidar_processing_amd import ClusteringConfiguration
want, wn = oracle.cluster(big[r["obstacle_idx"]], oracle.CluCfg(0.25, 0.5))
obs4 = np.ascontiguousarray(big[r["obstacle_idx"]])
for mode in ("search", "lists"):
    c = Context(0); c.set_neighbour_mode(mode); c.reserve(obs4.shape[0], 600)
    bad = 0
    for rep in range(int(sys.argv[1])):
        lab, nc = c.cluster(obs4, ClusteringConfiguration(0.25, 0.5))
        if nc != wn or not np.array_equal(lab, want):
            bad += 1
            if bad == 1:
                d = np.nonzero(lab != want)[0]
                print(mode, "first mismatch at rep", rep, "points differing", len(d), d[:8], lab[d[:8]], want[d[:8]])
    print(mode, "cluster reps", sys.argv[1], "mismatches", bad)
    c.close()
# ... and the FIRST call of a fresh context with the default list workspace (64 + 192 words per point): groups that find no
# single-pass room count their lists and take the exact-length region -- the path a long-lived context grows out of
fresh = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
stat = []
for rep in range(fresh):
    c = Context(0); c.set_neighbour_mode("lists")
    try:
        lab, nc = c.cluster(obs4, ClusteringConfiguration(0.25, 0.5))
        st = c.frame_stats()
        stat.append((st["neighbour_entries"], st["neighbour_words"]))
        if nc != wn or not np.array_equal(lab, want):
            bad += 1
            d = np.nonzero(lab != want)[0]
            print("fresh lists context rep", rep, "points differing", len(d), d[:8], lab[d[:8]], want[d[:8]], "clusters", nc, wn)
    finally:
        c.close()
print("fresh lists contexts", fresh, "mismatches", bad, "entries / words of the last", stat[-1])
