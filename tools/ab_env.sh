#!/bin/bash
# bench with an environment variable set to each of the given values (development tool): tools/ab_env.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift 2
for v in $vals; do
  for m in "" "--cloudy"; do
    for rep in 1 2; do
      env $var=$v python bench.py --no-cpu-baseline --no-extra $m "$@" 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$var=$v %-8s %9d col/s  %.4f ms  median %.4f' % ('$m', j['value'], j['ms_per_step'], j['config']['ms_per_step_median']))"
    done
  done
done
