cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/dsprof; rm -rf $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out -- python -c "
import sys; sys.path.insert(0,'tools/micro'); import logging; logging.disable(logging.WARNING)
import device_step_timing as d
print(d.loop(True, 300, False))" > $out.log 2>&1
f=$(find $out -name "*.db" | head -1); python tools/rocpd_stats.py $f > gpurun_out/dsprof.txt; rm -rf $out; tail -2 $out.log; head -24 gpurun_out/dsprof.txt
