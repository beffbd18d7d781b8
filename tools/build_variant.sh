#!/bin/bash
# A/B variant of librrtmg_hip.so (development tool): recompiles ONE translation unit with extra -D switches and links it
# with the product's other objects.   tools/build_variant.sh <name> <sw|lw> "<flags>"  ->  climt_amd/_lib/ab/libv_<name>.so (travels with gpurun while it exists; delete after the session)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; tu=$2; flags=$3
prod=climt_amd/_lib/obj-da39a3ee
mkdir -p climt_amd/_lib/var climt_amd/_lib/ab
obj=climt_amd/_lib/var/${name}_rrtmg_${tu}.hip.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c climt_amd/csrc/rrtmg_${tu}.hip -o $obj 2> climt_amd/_lib/var/${name}.log || { cat climt_amd/_lib/var/${name}.log; exit 1; }
others=$(ls $prod/*.o | grep -v "rrtmg_${tu}.hip.o")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o climt_amd/_lib/ab/libv_${name}.so $obj $others
echo "built libv_${name}.so ($flags)"
