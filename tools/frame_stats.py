import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from lidar_processing_amd import Context, SegmentationConfiguration, ClusteringConfiguration
from util import FRAMES, load_frame
ctx = Context(0)
for f in FRAMES:
    r = ctx.segment_cluster(load_frame(f), SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5), ClusteringConfiguration(distance_squared=0.25, cluster_quality=0.5))
    print(f, ctx.frame_stats())
