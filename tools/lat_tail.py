"""where the slow iterations of a one-frame closed loop are (device-resident, lists mode)"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import load_frame, FRAMES
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
pts = load_frame(FRAMES[0]); n = pts.shape[0]
c = Context(0); c.reserve(n)
rec = np.zeros((n, 8), np.float32); rec[:, :4] = pts
d = torch.from_numpy(rec).cuda()
out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
pl = torch.empty(24, dtype=torch.float32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
def run():
    c.segment_cluster_device(d.data_ptr(), 32, n, scfg, ccfg, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), pl.data_ptr(), out[3].data_ptr(), cnt.data_ptr()); c.synchronize()
ts = []
for i in range(600):
    a = time.perf_counter(); run(); ts.append((time.perf_counter() - a) * 1e3)
ts = np.array(ts)
print("first 8:", np.round(ts[:8], 2)); print("p50 %.3f p90 %.3f p99 %.3f max %.3f" % (np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max()))
slow = np.nonzero(ts > 2 * np.median(ts))[0]; print("slow iterations:", slow[:40], np.round(ts[slow][:40], 1))
