#!/bin/bash
# interleaved A/B of library variants (development tool): tools/ab_libs.sh "libA.so libB.so ..." [reps] [bench args] -- one line per run
libs=$1; reps=${2:-3}; shift 2
for rep in $(seq $reps); do
  for m in "" "--cloudy"; do
    for lib in $libs; do
      L=$PWD/climt_amd/_lib/$lib; [ $lib = product ] && L=$PWD/climt_amd/_lib/librrtmg_hip.so
      RRTMG_HIP_LIB=$L python bench.py --no-cpu-baseline --no-extra $m "$@" 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-14s %-8s %9d col/s  %.4f ms  median %.4f  sw %.3f lw %.3f' % ('$lib', '$m', j['value'], j['ms_per_step'], j['config']['ms_per_step_median'], r['sw_solve_ms'], r['lw_solve_ms']))"
    done
  done
done
