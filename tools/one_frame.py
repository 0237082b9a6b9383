import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import load_frame
pts = load_frame(sys.argv[1] if len(sys.argv) > 1 else "0000000000")
ctx = Context(0)
for _ in range(3):
    out = ctx.segment_cluster(pts, SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5), ClusteringConfiguration(0.25, 0.5))
print(out["n_clusters"])
