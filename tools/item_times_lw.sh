#!/bin/bash
# lone-item durations of the longwave kernel for library variants (development tool): tools/item_times_lw.sh lib1 lib2 ...
for lib in "$@"; do
  L=$PWD/climt_amd/_lib/$lib; [ $lib = product ] && L=$PWD/climt_amd/_lib/librrtmg_hip.so
  RRTMG_HIP_LIB=$L python tools/item_times.py 8192 2>&1 | grep -A 14 "^lw clear" | awk -v l=$lib 'NR==1 {print l, $0} NR>1 {printf "%s ", $NF=="ms" ? $(NF-1) : $NF} END {print ""}'
done
