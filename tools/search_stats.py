"""expansion-driven search counters per frame (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame
ctx = Context(0)
for f in FRAMES:
    out = ctx.segment_cluster(load_frame(f), SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5),
                              ClusteringConfiguration(0.25, 0.5))
    st = ctx.frame_stats()
    print(f, {k: st[k] for k in ("n_obstacle", "components", "expansions", "windows", "replay_entries", "candidates", "overflows")},
          "exp/window %.2f cand/exp %.0f hits/exp %.1f" % (st["expansions"] / max(1, st["windows"]), st["candidates"] / max(1, st["expansions"]),
                                                          st["replay_entries"] / max(1, st["expansions"])))
