#!/bin/bash
# tools/r4_probe17.sh: the look-ahead of the two-call form (lpx_set_lookahead) -- its tests, then the two calls of the
# unchanged node through the C++ headers with and without it
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p17; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "looks_ahead or recognises or dropin" 2>&1 | tail -5
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from tests import util
sys.path.insert(0, 'tests')
import util
wl = bench.WORKLOADS['stream']
f = util.load_stream_frame(util.stream_names()[0])
for k in range(2):
    d = bench.dropin_cxx_latency(f, wl)
    print(json.dumps({k: v for k, v in d.items() if k != 'what'}))
PY
