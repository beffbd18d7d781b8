#!/bin/bash
# Profiles of the bench command itself (run on the GPU box through gpurun); summaries land in gpurun_out/ and the
# ones to be judged are copied into profiles/ afterwards.  Kernel trace and each PMC counter are separate passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r01}
for mode in clear cloudy; do
  flag=""; [ $mode = cloudy ] && flag="--cloudy"
  for variant in overlap serial; do
    vf=""; [ $variant = serial ] && vf="--serial"
    out=gpurun_out/prof_${mode}_$variant; rm -rf $out
    timeout 240 rocprofv3 --kernel-trace --stats -d $out -- python bench.py --no-cpu-baseline $flag $vf > $out.log 2>&1
    grep "^{" $out.log | tail -1 > gpurun_out/${R}_bench_${mode}_$variant.json
    f=$(find $out -name "*.db" | head -1)
    [ -n "$f" ] && python tools/rocpd_stats.py $f > gpurun_out/${R}_bench_${mode}_${variant}_kernel_stats.txt
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc_${mode}_$c; rm -rf $out
    timeout 240 rocprofv3 --kernel-trace --pmc $c -d $out -- python bench.py --no-cpu-baseline --serial --steps 3 --warmup 1 $flag > $out.log 2>&1
    f=$(find $out -name "*.db" | head -1)
    [ -n "$f" ] && python tools/rocpd_pmc.py $f > gpurun_out/${R}_pmc_${mode}_$c.txt
  done
done
ls -la gpurun_out/${R}_*
