#!/bin/bash
# A/B over problem sizes (development tool): tools/ab_sizes.sh "8192 20000 32768" lib1.so [lib2.so ...]
sizes="$1"; shift
for lib in "$@"; do
  for n in $sizes; do
    for m in "" "--cloudy"; do
      RRTMG_HIP_LIB=$PWD/climt_amd/_lib/$lib timeout 300 python bench.py --no-cpu-baseline --columns $n $m 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-20s %6d %-9s %8d col/s %7.3f ms  sw %.3f lw %.3f (serial kernel ms)' % ('$lib', $n, '$m', j['value'], j['ms_per_step'], r['sw_solve_ms_serial'], r['lw_solve_ms_serial']))"
    done
  done
done
