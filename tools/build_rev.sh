#!/bin/bash
# tools/build_rev.sh REV NAME [extra hipcc flags...] -- builds the library sources AS COMMITTED at git revision REV into
# lidar_processing_amd/ab/liblpx_NAME.so (release flags, no -DLPX_DEV_KNOBS unless passed), for an A/B of the working
# tree against an earlier commit on ONE GPU box: LPX_LIB=lidar_processing_amd/ab/liblpx_NAME.so python bench.py ...
# (tools/probe.sh ab OUT WORKLOAD "NAME default").  Runs here (the GPU box has no .git); the built .so travels.
set -e
rev=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=/tmp/lpx_rev_$name
rm -rf "$src"; mkdir -p "$src" "$root/lidar_processing_amd/ab"
git -C "$root" archive "$rev" lidar_processing_amd/csrc include $(git -C "$root" ls-tree --name-only "$rev" experiments 2>/dev/null) | tar -x -C "$src"
cd "$src/lidar_processing_amd/csrc"
pids=()
for f in $(ls *.hip | sed "s/\.hip$//"); do  # (every source of the tree as it is / was: the file set changed in round 6)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. \
    -Wno-unused-value -Wno-unused-result "$@" -c $f.hip -o $f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/lidar_processing_amd/ab/liblpx_$name.so" *.o
echo "built lidar_processing_amd/ab/liblpx_$name.so from $rev"
