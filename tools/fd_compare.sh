# development: the one-rank distributed code path of bench.py against the plain single-GPU loop (same kernels, same streams)
export RRTMG_HIP_ALLOW_SYNTHETIC_LW=1
F="--steps 200 --no-extra --no-cpu-baseline"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), d['config'].get('communicator'), round(d['roofline']['sw_solve_ms'],3), round(d['roofline']['lw_solve_ms'],3))
"; }
timeout 200 python bench.py $F 2>/dev/null | show plain
GPU_MAX_HW_QUEUES=4 timeout 200 python bench.py $F 2>/dev/null | show plain-4q
timeout 200 python bench.py $F --cloudy 2>/dev/null | show plain-cloudy
GPU_MAX_HW_QUEUES=4 timeout 200 python bench.py $F --cloudy 2>/dev/null | show plain-cloudy-4q
for g in none all root; do timeout 200 python bench.py $F --force-dist --gather $g 2>/dev/null | show fd-$g; done
timeout 200 python bench.py $F --force-dist --gather all --comm torch 2>/dev/null | show fd-all-torch
timeout 200 python bench.py $F --force-dist --gather all --config 4 2>/dev/null | show fd-all-config4
timeout 200 python bench.py $F --config 4 2>/dev/null | show plain-config4
