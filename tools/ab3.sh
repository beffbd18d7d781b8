#!/bin/bash
# serial-mode kernel timing of library variants (development tool; variants may compute wrong results)
for lib in "$@"; do
    RRTMG_HIP_LIB=$PWD/climt_amd/_lib/$lib timeout 300 python bench.py --no-cpu-baseline --serial --steps 10 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-34s %8d col/s %7.3f ms  sw %.3f lw %.3f' % ('$lib', j['value'], j['ms_per_step'], r['sw_solve_ms_serial'], r['lw_solve_ms_serial']))"
done
