#!/bin/bash
# tools/r4_probe21.sh: the launch-shape knobs of round 3 (kd tail staging, workgroup sizes of the kd levels, workgroups of
# the cell linking, replay workgroups, kd group size) once more under the round-4 kernels, headline shape, one box
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=$GRAFT_REPO_ROOT/gpurun_out/r4p21; mkdir -p $O
export LPX_LIB=$GRAFT_REPO_ROOT/lidar_processing_amd/liblpx_dev.so
B="--workload stream --no-cpu-baseline --no-latency --no-inflight --no-sub --no-verify"
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $B --steps 6 --warmup 2 --contexts 16 --frames-per-step 1024 2>$O/$name.err | tail -1 > $O/$name.json
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['completion']['p99_frame_completion_ms'])"
}
run base1 X=1
run tail512 LPX_KD_TAIL=512
run tail1984 LPX_KD_TAIL=1984
run wide40k LPX_KD_WIDE=40000
run wide200k LPX_KD_WIDE=200000
run gp16_64 LPX_GP_G0=16 LPX_GP_G1=64
run gp64_256 LPX_GP_G0=64 LPX_GP_G1=256
run base2 X=1
run rs4 LPX_RS_GRID=4
run rs2 LPX_RS_GRID=2
run rs1 LPX_RS_GRID=1
run bucket32 LPX_IX_BUCKET=32
run spine0 LPX_IX_SPINE=0
run remapff LPX_REMAP=ff
run remap3f LPX_REMAP=3f
run base3 X=1
