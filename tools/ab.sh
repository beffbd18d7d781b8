#!/bin/bash
# A/B bench of library variants: tools/ab.sh libA.so libB.so ...   (development tool)
for lib in "$@"; do
  for mode in "" "--cloudy"; do
    RRTMG_HIP_LIB=$PWD/climt_amd/_lib/$lib python bench.py --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib $mode', round(j['value']), 'col/s', round(j['ms_per_step'],3), 'ms  sw', round(r['sw_solve_ms'],3), 'lw', round(r['lw_solve_ms'],3))"
  done
done
