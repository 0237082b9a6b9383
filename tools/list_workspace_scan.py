import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import load_frame, FRAMES
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
for nb, rs in [(256, 512), (64, 192), (64, 160), (64, 128), (48, 208)]:
    res = []
    for f in FRAMES:
        pts = load_frame(f); n = pts.shape[0]
        c = Context(0); c.reserve(n, nb); c.reserve_single_pass(rs)
        rec = np.zeros((n, 8), np.float32); rec[:, :4] = pts
        d = torch.from_numpy(rec).cuda()
        out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
        pl = torch.empty(24, dtype=torch.float32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        def run():
            c.segment_cluster_device(d.data_ptr(), 32, n, scfg, ccfg, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), pl.data_ptr(), out[3].data_ptr(), cnt.data_ptr()); c.synchronize()
        for _ in range(5): run()
        ts = []
        for _ in range(30):
            a = time.perf_counter(); run(); ts.append(time.perf_counter() - a)
        st = c.frame_stats()
        res.append((round(float(np.median(ts)) * 1e3, 3), int(cnt.cpu()[3]), st["neighbour_entries"], st["neighbour_words"]))
        c.close()
    print(nb, rs, res, "MB per 123k frame:", round((nb + rs) * 4 * 123398 / 1e6))
