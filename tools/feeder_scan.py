"""PCIe-inclusive feeder rates (lpx_feeder_run_multi) over lane counts; env knobs come from the caller.
usage: feeder_scan.py "1 4 8 20" [batch] [frames]"""
import os, sys, time, tempfile
if not os.environ.get("LPX_SCAN_NO_HWQ"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")  # like bench.py: streams that share a hardware queue serialise
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lidar_processing_amd as lpx
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import load_stream_frame, stream_names
lanes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1 4 8 20").split()]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
frames = [load_stream_frame(n) for n in stream_names()]
scfg = SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5); ccfg = ClusteringConfiguration(0.25, 0.5)
with tempfile.TemporaryDirectory() as tmp:
    paths = []
    for j, hf in enumerate(frames):
        paths.append(os.path.join(tmp, f"{j:010d}.pcd")); lpx.write_pcd(paths[-1], hf)
    feeder = lpx.Feeder(paths, 0)
    ids = np.array([i % len(frames) for i in range(NF)], np.uint32)
    pts = sum(frames[i].shape[0] for i in ids.tolist())
    for C in lanes:
        ctxs = [Context(0, batch=B) for _ in range(C)]
        for c in ctxs:
            c.reserve(max(f.shape[0] for f in frames))
        out = feeder.run(ctxs, ids, scfg, ccfg)
        a = time.perf_counter()
        for _ in range(3):
            feeder.run(ctxs, ids, scfg, ccfg, out)
        t = (time.perf_counter() - a) / 3
        print({k: os.environ[k] for k in os.environ if k.startswith(("LPX_", "HSA_ENABLE_SDMA", "GPU_MAX"))}, "lanes", C, "batch", B, "frames", NF,
              "frames/s %.0f" % (len(ids) / t), "Mpts/s %.0f" % (pts / t / 1e6), "H2D GB/s %.1f" % (pts * 16 / t / 1e9), flush=True)
        for c in ctxs:
            c.close()
        del out
    feeder.close()
