#!/bin/sh
# Binds liblpx.so into a checkout of YevgeniyEngineer/LiDAR-Processing.
#
#   integration/apply_to_reference.sh <reference checkout> [<this repository>]
#
# src/processor.cpp includes "segmentation.hpp" and "clustering.hpp" with QUOTES, and a quoted include looks in the
# directory of the including file first -- before every -I / target_include_directories path.  Adding an include
# path is therefore not enough: as long as src/segmentation.hpp and src/clustering.hpp exist, the CPU classes
# are what gets compiled.  This script
#   1. keeps the originals as src/*_cpu.hpp / src/*_cpu.cpp (nothing is deleted),
#   2. puts one-line forwarding headers in their place, so every translation unit of the node (processor.cpp,
#      polygonization.hpp) picks up the drop-in classes without a source change,
#   3. prints the three CMake lines to change.
set -eu
REF=${1:?usage: apply_to_reference.sh <reference checkout> [<this repository>]}
LPX=${2:-$(cd "$(dirname "$0")/.." && pwd)}
for h in segmentation clustering; do
    [ -f "$REF/src/$h.hpp" ] || { echo "no $REF/src/$h.hpp" >&2; exit 1; }
    if ! grep -q "lpx drop-in forwarding header" "$REF/src/$h.hpp"; then
        mv "$REF/src/$h.hpp" "$REF/src/${h}_cpu.hpp"
        [ -f "$REF/src/$h.cpp" ] && mv "$REF/src/$h.cpp" "$REF/src/${h}_cpu.cpp"
    fi
    printf '// lpx drop-in forwarding header (the CPU original is %s_cpu.hpp)\n#include "%s/include/lidar_processing/%s.hpp"\n' \
        "$h" "$LPX" "$h" > "$REF/src/$h.hpp"
done
cat <<MSG
forwarding headers written to $REF/src/{segmentation,clustering}.hpp
now, in $REF/CMakeLists.txt:
  * remove src/segmentation.cpp and src/clustering.cpp from add_executable(processor ...)
  * target_include_directories(processor PRIVATE $LPX/include)
  * target_link_libraries(processor $LPX/lidar_processing_amd/liblpx.so)      # Eigen3 / TBB are no longer needed
MSG
