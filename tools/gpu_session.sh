#!/bin/bash
# One gpurun session (development tool): tools/gpu_session.sh <tag> <sections...>
#   sections: tests smoke bench dist sizes ab:<lib1,lib2,...> phases prof profsizes pmc pmclarge sq
# Everything lands under gpurun_out/<tag>_*; the summaries to be judged are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RRTMG_HIP_ALLOW_SYNTHETIC_LW=1
R=$1; shift
O=gpurun_out
mkdir -p $O
# every summary starts with the hash of the sources the profiled library was built from (climt_amd/build.py::source_hash):
# bench.py compares it with the library that runs before it quotes the counters
HASH=$(python -c "from climt_amd._lib import source_hash; print(source_hash())" 2>/dev/null | tail -1)
stats() { f=$(find $1 -name "*.db" | head -1); [ -n "$f" ] && { echo "# source_hash $HASH"; python tools/rocpd_stats.py $f; } > $2; rm -rf $1; }   # (the raw traces stay on the box: gpurun_out/ is capped at 64 MiB)
pmc() { f=$(find $1 -name "*.db" | head -1); [ -n "$f" ] && { echo "# source_hash $HASH"; python tools/rocpd_pmc.py $f rrtmg::; } > $2; rm -rf $1; }
for sec in "$@"; do
  echo "=== section $sec ($(date +%T))"
  case $sec in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/${R}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${R}_pytest.log; tail -25 $O/${R}_pytest.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    bench)
      timeout 600 python bench.py > $O/${R}_bench_default.log 2>&1; grep "^{" $O/${R}_bench_default.log | tail -1 > $O/${R}_bench_default.json; cut -c1-600 $O/${R}_bench_default.json
      timeout 600 python bench.py --cloudy > $O/${R}_bench_default_cloudy.log 2>&1; grep "^{" $O/${R}_bench_default_cloudy.log | tail -1 > $O/${R}_bench_default_cloudy.json; cut -c1-400 $O/${R}_bench_default_cloudy.json ;;
    dist)
      # ONE run measures every gather mode (bench.py: gather_modes); the second line: two ranks on this one GPU over torch/gloo (testing communicator)
      timeout 400 python bench.py --force-dist --no-cpu-baseline --no-extra --steps 100 > $O/${R}_dist_modes.log 2>&1; echo "dist rc=$?"; grep "^{" $O/${R}_dist_modes.log | tail -1 > $O/${R}_dist_modes.json
      python -c "
import json
j=json.load(open('$O/${R}_dist_modes.json')); print('headline', j['config']['gather_mode'], round(j['value']), j['ms_per_step'], j['config']['communicator'])
for m,g in j['gather_modes'].items():
    if m[0] != '_': print('  %-7s %9d col/s %7.3f ms  ingress %.1f MB/step  achieved %.1f GB/s  ran=%s err=%s' % (m, g['value'], g['ms_per_step'], g['ingress_bytes_per_gpu_per_step']/1e6, g['ingress_GBps_per_gpu_achieved'], g['gather_ran'], g['error']))" || tail -5 $O/${R}_dist_modes.log
      timeout 300 python bench.py --gpus 2 --share-device --dist-backend gloo --comm torch --no-cpu-baseline --no-extra --steps 50 --min-seconds 1 > $O/${R}_dist_torch.log 2>&1; grep "^{" $O/${R}_dist_torch.log | tail -1 | cut -c1-300 ;;
    sizes)
      for c in 4 5; do
        timeout 600 python bench.py --config $c --no-cpu-baseline --no-extra > $O/${R}_bench_config$c.log 2>&1; grep "^{" $O/${R}_bench_config$c.log | tail -1 > $O/${R}_bench_config$c.json; cut -c1-300 $O/${R}_bench_config$c.json
      done
      timeout 600 python bench.py --columns 131072 --no-cpu-baseline --no-extra > $O/${R}_bench_clear131072.log 2>&1; grep "^{" $O/${R}_bench_clear131072.log | tail -1 > $O/${R}_bench_clear131072.json; cut -c1-300 $O/${R}_bench_clear131072.json ;;
    ab:*)
      # entries: <lib>[+ENV=VALUE[+ENV=VALUE...]]; AB_MODES="clear" | "cloudy" | "clear cloudy" (default both)
      for ent in $(echo ${sec#ab:} | tr , ' '); do
        lib=${ent%%+*}; envs=""; [ "$ent" != "$lib" ] && envs=$(echo ${ent#*+} | tr + ' ')
        for m in ${AB_MODES:-clear cloudy}; do
          mode=""; [ $m = cloudy ] && mode="--cloudy"
          L=$PWD/climt_amd/_lib/ab/$lib; [ $lib = product ] && L=$PWD/climt_amd/_lib/librrtmg_hip.so
          env $envs RRTMG_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-extra --no-mcica --steps 150 $mode 2>&1 | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); r=j['roofline']
    print('%-40s %-8s %9d col/s %7.3f ms (median %.3f)  sw %.3f lw %.3f | serial sw %.3f lw %.3f' % ('$ent', '$mode', j['value'], j['ms_per_step'], j['config']['ms_per_step_median'], r['sw_solve_ms'], r['lw_solve_ms'], r['sw_solve_ms_serial'], r['lw_solve_ms_serial']))
except Exception as e:
    print('$lib $mode FAILED', e)" | tee -a $O/${R}_ab.txt
        done
      done ;;
    absize:*)
      # as ab:, on the large grids: AB_SIZES = ';'-separated bench.py argument sets (default: configs 4 and 5 shards, 131 072 clear-sky and McICA columns)
      IFS=';' read -ra SPECS <<< "${AB_SIZES:---config 4;--config 5;--columns 131072;--columns 131072 --cloudy}"
      for rep in 1 2; do
      for spec in "${SPECS[@]}"; do
        for ent in $(echo ${sec#absize:} | tr , ' '); do
          lib=${ent%%+*}; envs=""; [ "$ent" != "$lib" ] && envs=$(echo ${ent#*+} | tr + ' ')
          L=$PWD/climt_amd/_lib/ab/$lib; [ $lib = product ] && L=$PWD/climt_amd/_lib/librrtmg_hip.so
          env $envs RRTMG_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extra --no-mcica --min-seconds 1.5 $spec 2>&1 | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); r=j['roofline']
    print('%-44s %-28s %9d col/s %8.3f ms  sw %.3f lw %.3f per launch' % ('$ent', '$spec', j['value'], j['ms_per_step'], r['sw_solve_ms'], r['lw_solve_ms']))
except Exception as e:
    print('$ent $spec FAILED', e)" | tee -a $O/${R}_absize.txt
        done
      done
      done ;;
    pmcab:*)
      # a compact counter set on library variants, clear sky, the kernels alone: pmcab:<lib>,<lib>  (PMC_FILTER: kernel-name substring)
      for lib in $(echo ${sec#pmcab:} | tr , ' '); do
        L=$PWD/climt_amd/_lib/ab/$lib; [ $lib = product ] && L=$PWD/climt_amd/_lib/librrtmg_hip.so
        i=0
        for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
                 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
                 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
                 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
                 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
          i=$((i+1)); out=$O/pmcab_$i; rm -rf $out
          RRTMG_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc $P -d $out -- python bench.py --no-cpu-baseline --no-extra --no-mcica --serial --steps 3 --warmup 1 --min-seconds 0 > $out.log 2>&1
          f=$(find $out -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f ${PMC_FILTER:-lw_solve_all_kernel} | sed "s/^/$lib /" >> $O/${R}_pmcab.txt; rm -rf $out
        done
      done; cat $O/${R}_pmcab.txt ;;
    phases)
      # the diagnostic build with phase timers in the longwave sweeps (climt_amd/_lib/lib_profile.so: RRTMG_HIP_BUILD_FLAGS=-DRRTMG_PROFILE RRTMG_HIP_BUILD_OUT=.../lib_profile.so python climt_amd/build.py --force)
      for mode in "" "--cloudy"; do
        RRTMG_HIP_LIB=$PWD/climt_amd/_lib/lib_profile.so timeout 200 python bench.py --no-cpu-baseline --no-extra --serial --steps 2 --warmup 1 --min-seconds 0 $mode 2>&1 | grep -A1 "lw phases" | tail -2 | tee -a $O/${R}_lw_phases.txt
      done ;;
    prof)
      for mode in clear cloudy; do
        flag=""; [ $mode = cloudy ] && flag="--cloudy"
        for variant in overlap serial; do
          vf=""; [ $variant = serial ] && vf="--serial"
          out=$O/prof_${mode}_$variant; rm -rf $out
          timeout 240 rocprofv3 --kernel-trace --stats -d $out -- python bench.py --no-cpu-baseline --no-extra --no-mcica --steps 40 --warmup 3 $flag $vf > $out.log 2>&1
          grep "^{" $out.log | tail -1 > $O/${R}_bench_${mode}_$variant.json
          stats $out $O/${R}_bench_${mode}_${variant}_kernel_stats.txt
        done
      done; head -12 $O/${R}_bench_clear_serial_kernel_stats.txt ;;
    profsizes)
      # kernel stats of the large configurations (BASELINE configs 4 and 5 at shard size, 131 072 clear-sky columns)
      for spec in "config4:--config 4 --steps 12" "config5:--config 5 --steps 3" "clear131072:--columns 131072 --steps 6"; do
        tag=${spec%%:*}; args=${spec#*:}
        out=$O/prof_$tag; rm -rf $out
        timeout 400 rocprofv3 --kernel-trace --stats -d $out -- python bench.py --no-cpu-baseline --no-extra --no-mcica --warmup 2 --min-seconds 0 $args > $out.log 2>&1
        grep "^{" $out.log | tail -1 > $O/${R}_bench_${tag}_profiled.json
        stats $out $O/${R}_bench_${tag}_kernel_stats.txt; head -8 $O/${R}_bench_${tag}_kernel_stats.txt
      done ;;
    pmc|pmclarge)
      n=8192; tag=""; [ $sec = pmclarge ] && { n=131072; tag="_131072"; }
      for mode in clear cloudy; do
        flag=""; [ $mode = cloudy ] && flag="--cloudy"
        for c in FETCH_SIZE WRITE_SIZE; do
          out=$O/pmc_${mode}_$c$tag; rm -rf $out
          timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out -- python bench.py --columns $n --no-cpu-baseline --no-extra --no-mcica --serial --steps 3 --warmup 1 --min-seconds 0 $flag > $out.log 2>&1
          pmc $out $O/${R}_pmc_${mode}_$c$tag.txt; cat $O/${R}_pmc_${mode}_$c$tag.txt
        done
      done ;;
    sq)
      for mode in clear cloudy; do
        flag=""; [ $mode = cloudy ] && flag="--cloudy"
        i=0
        for P in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
          i=$((i+1)); out=$O/sq_${mode}_$i; rm -rf $out
          timeout 300 rocprofv3 --kernel-trace --pmc $P -d $out -- python bench.py --no-cpu-baseline --no-extra --no-mcica --serial --steps 3 --warmup 1 --min-seconds 0 $flag > $out.log 2>&1
          pmc $out $O/${R}_sq_${mode}_$i.txt
        done
        cat $O/${R}_sq_${mode}_1.txt $O/${R}_sq_${mode}_2.txt > $O/${R}_pmc_${mode}_sq.txt
      done; cat $O/${R}_pmc_clear_sq.txt ;;
  esac
done
echo "=== done ($(date +%T))"
