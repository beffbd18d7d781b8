#!/bin/bash
# A/B bench (development tool): tools/ab2.sh lib1.so [lib2.so ...]; prints overlap/serial/cloudy throughput and serial kernel ms
for lib in "$@"; do
  for m in "--serial" "" "--cloudy"; do
    RRTMG_HIP_LIB=$PWD/climt_amd/_lib/$lib timeout 300 python bench.py --no-cpu-baseline $m 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-28s %-9s %8d col/s %7.3f ms  sw %.3f lw %.3f (serial kernel ms)' % ('$lib', '$m', j['value'], j['ms_per_step'], r['sw_solve_ms_serial'], r['lw_solve_ms_serial']))"
  done
done
