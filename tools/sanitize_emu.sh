#!/usr/bin/env bash
# Development tool (CPU only): the host emulation of the device functions -- the same __host__ __device__ sources the kernels run
# (climt_amd/csrc/rrtmg_{sw,lw}_device.h through tests/emu/) -- built with AddressSanitizer + UndefinedBehaviorSanitizer, and the CPU
# suite run on it.  GPU sanitizers are not available on the pool; this is where an out-of-range table index or a signed overflow in
# the physics shows up.  The sanitized library replaces tests/_emu/librrtmg_emu.so for the run and is rebuilt normally afterwards.
#   tools/sanitize_emu.sh            (-O0 for the two emulation units: the optimiser needs > 10 min per unit with the sanitizers on)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
W=$(mktemp -d)
CC="hipcc --offload-arch=gfx950 --cuda-host-only -O0 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-sanitize=vptr -fno-omit-frame-pointer -shared-libsan"
for src in "$ROOT/tests/emu/emu_sw.hip" "$ROOT/tests/emu/emu_lw.hip" "$ROOT/climt_amd/csrc/rrtmg_tables.cpp" "$ROOT/tests/emu/mt_host_stream.cpp"; do
  $CC -c "$src" -o "$W/$(basename "$src").o" &
done
wait
$CC -shared -o "$W/librrtmg_emu.so" "$W"/*.o
cp "$W/librrtmg_emu.so" "$ROOT/tests/_emu/librrtmg_emu.so"
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
rc=0
(cd "$ROOT" && LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
   python -m pytest tests -x -q -m "not gpu" "$@") || rc=$?
"$ROOT/tests/emu/build.sh" > /dev/null
rm -rf "$W"
exit $rc
