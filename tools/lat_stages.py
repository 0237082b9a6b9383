"""stage times (HIP events, lpx_profile_*) of ONE frame on a single-frame context, lists and search mode"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import load_frame, FRAMES
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
f = FRAMES[0]
pts = load_frame(f); n = pts.shape[0]
c = Context(0); c.reserve(n)
rec = np.zeros((n, 8), np.float32); rec[:, :4] = pts
d = torch.from_numpy(rec).cuda()
out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
pl = torch.empty(24, dtype=torch.float32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
def run():
    c.segment_cluster_device(d.data_ptr(), 32, n, scfg, ccfg, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), pl.data_ptr(), out[3].data_ptr(), cnt.data_ptr()); c.synchronize()
for _ in range(5): run()
ts = []
for _ in range(40):
    a = time.perf_counter(); run(); ts.append(time.perf_counter() - a)
print("median ms", round(float(np.median(ts)) * 1e3, 3))
c.profile_enable(True)
for _ in range(20): run()
st = c.profile_read()
tot = 0.0
for k, (ms, k_cnt) in st.items():
    if k_cnt:
        print("%-14s %8.4f ms  (%d)" % (k, ms / k_cnt, k_cnt)); tot += ms / k_cnt
print("sum of stages", round(tot, 4))
