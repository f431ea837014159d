# where do the non-MFMA cycles of the NT kernel go?  PMC passes (one group per run) over benchmarks/gemm_pmc_probe.py: our kernel next to hipBLASLt's
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
ARGS="${GEMM_ARGS:-8192 3072 768 32}"
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "GRBM_GUI_ACTIVE MfmaUtil"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/gemm_pmc/p$i -- python $R/benchmarks/gemm_pmc_probe.py $ARGS > $R/gpurun_out/gemm_pmc_p$i.log 2>&1 || tail -3 $R/gpurun_out/gemm_pmc_p$i.log
done
cd $R
python benchmarks/pmc_by_kernel.py "sgemm_nt_asm|sgemm_nt16|Cijk" gpurun_out/gemm_pmc/p* > gpurun_out/gemm_pmc_summary.txt
rm -rf gpurun_out/gemm_pmc
cat gpurun_out/gemm_pmc_summary.txt
