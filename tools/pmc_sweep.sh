#!/bin/bash
# PMC sweep of the solve kernels (development tool): tools/pmc_sweep.sh [clear|cloudy] [ncol]
# Each counter group is its own rocprofv3 pass (kernel-trace + pmc only), each under its own timeout.
mode=${1:-clear}; n=${2:-8192}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r P; do
  i=$((i+1)); out=gpurun_out/pmcS$i; rm -rf $out
  timeout 90 rocprofv3 --kernel-trace --pmc $P -d $out -- python tools/gpu_prof_run.py $n $mode 2 > $out.log 2>&1
  f=$(find $out -name "*.db" 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocpd_pmc.py $f _solve_ || { echo "pass $i failed: $P"; tail -3 $out.log; }
done <<'SETS'
SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
SETS
