#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- builds lidar_processing_amd/ab/liblpx_NAME.so with extra
# compile flags (objects under /tmp), for A/B runs on the GPU box: LPX_LIB=lidar_processing_amd/ab/liblpx_NAME.so.
# Development only (built with -DLPX_DEV_KNOBS like liblpx_dev.so: the LPX_* environment knobs are read): the product
# library is lidar_processing_amd/liblpx.so.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/lpx_variant_$name
rm -rf "$obj"; mkdir -p "$obj" "$root/lidar_processing_amd/ab"
cd "$root/lidar_processing_amd/csrc"
pids=()
for f in $(ls *.hip | sed "s/\.hip$//"); do  # (every source of the tree as it is / was: the file set changed in round 6)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. \
    -Wno-unused-value -Wno-unused-result -DLPX_DEV_KNOBS "$@" -c $f.hip -o "$obj/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/lidar_processing_amd/ab/liblpx_$name.so" "$obj"/*.o
echo "built lidar_processing_amd/ab/liblpx_$name.so"
