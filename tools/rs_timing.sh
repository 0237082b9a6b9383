#!/bin/sh
# diagnostic build of the replay with s_memtime stamps (LPX_RS_TIMING); restores the normal build afterwards
set -e
cd "$(dirname "$0")/.."
make -C lidar_processing_amd/csrc -s clean
make -C lidar_processing_amd/csrc -s -j8 CXXFLAGS_EXTRA=-DLPX_RS_TIMING
python - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from lidar_processing_amd import ClusteringConfiguration, Context, SegmentationConfiguration
from util import FRAMES, load_frame
ctx = Context(0)
for f in FRAMES:
    out = ctx.segment_cluster(load_frame(f), SegmentationConfiguration(number_of_planar_partitions=6, number_of_iterations=5), ClusteringConfiguration(0.25, 0.5))
    o = np.zeros(12, np.uint32); ctx._L.lpx_dbg_frame_stats_slot(ctx._h, 0, o.ctypes.data_as(C.c_void_p))
    s4 = np.zeros(4, np.uint32); ctx._L.lpx_dbg_search_stats_slot(ctx._h, 0, s4.ctypes.data_as(C.c_void_p))
    nb_entries = int(o[4]) | (int(o[5]) << 32)   # search cycles
    words = int(o[10]) | (int(o[11]) << 32)      # nb_total(+rs): gather cycles + apply... see below
    busiest = int(s4[0]) | (int(s4[1]) << 32)
    win = max(1, int(s4[2]))
    print(f, "windows", win, "| cycles per window: gather+select %.0f, search %.0f, apply %.0f | busiest workgroup %.3f ms at 2.4 GHz"
          % (words / win, nb_entries / win, int(s4[3]) * 1024 / win, busiest / 2.4e6))
PY
make -C lidar_processing_amd/csrc -s clean
make -C lidar_processing_amd/csrc -s -j8
