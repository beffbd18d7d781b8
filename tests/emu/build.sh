#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY: builds tests/_emu/librrtmg_emu.so (host emulation of the device functions)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$ROOT/tests/_emu"
SRC="$HERE/emu_sw.hip"
[ -f "$HERE/emu_lw.hip" ] && SRC="$SRC $HERE/emu_lw.hip"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -ffp-contract=off -o "$ROOT/tests/_emu/librrtmg_emu.so" \
  $SRC "$ROOT/climt_amd/csrc/rrtmg_tables.cpp" "$ROOT/climt_amd/csrc/rrtmg_mt.cpp"
echo "built tests/_emu/librrtmg_emu.so"
