cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gemm_pmc
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/gemm_pmc -- $R/scripts/ubench/gemm_i8 > $R/gpurun_out/gemm_pmc.log 2>&1
echo rc $?
find $R/gpurun_out/gemm_pmc -name "*.csv" | head
