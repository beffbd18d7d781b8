#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY: builds tests/_emu/librrtmg_emu.so (host emulation of the device functions).
# The two emulation units are compiled concurrently, then linked.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/tests/_emu"
mkdir -p "$OUT"
rm -f "$OUT"/*.o      # (objects of sources that have since moved must not be linked again)
CC="hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -ffp-contract=off"
pids=()
for src in "$HERE/emu_sw.hip" "$HERE/emu_lw.hip" "$ROOT/climt_amd/csrc/rrtmg_tables.cpp" "$HERE/mt_host_stream.cpp"; do
  $CC -c "$src" -o "$OUT/$(basename "$src").o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$CC -shared -o "$OUT/librrtmg_emu.so" "$OUT"/*.o
echo "built tests/_emu/librrtmg_emu.so"
