#!/bin/bash
# PMC sweep of the solve kernels (development tool): tools/pmc_sweep.sh [clear|cloudy] [ncol]
# Each counter group is its own rocprofv3 pass (kernel-trace + pmc only).
mode=${1:-clear}; n=${2:-8192}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r P; do
  i=$((i+1)); out=gpurun_out/pmcS$i; rm -rf $out
  rocprofv3 --kernel-trace --pmc $P -d $out -- python tools/gpu_prof_run.py $n $mode 2 > $out.log 2>&1
  f=$(find $out -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_pmc.py $f solve_all || { echo "pass $i failed: $P"; tail -3 $out.log; }
done <<'SETS'
SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TCC_WRITE_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum
SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES
SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED
SETS
