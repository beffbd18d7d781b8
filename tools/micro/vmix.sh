cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RRTMG_HIP_ALLOW_SYNTHETIC_LW=1
for mode in "" "--cloudy"; do
out=gpurun_out/vmix; rm -rf $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $out -- python bench.py --no-cpu-baseline --no-extra --no-mcica --serial --steps 3 --warmup 1 --min-seconds 0 $mode > $out.log 2>&1
f=$(find $out -name "*.db" | head -1); python tools/rocpd_pmc.py $f solve_; rm -rf $out
done
