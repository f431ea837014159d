R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
ARGS="${ATTN_ARGS:-128 64 64 12 64}"
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "GRBM_GUI_ACTIVE MfmaUtil" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/attn_pmc/p$i -- python $R/benchmarks/attn_pmc_probe.py $ARGS > $R/gpurun_out/attn_pmc_p$i.log 2>&1 || tail -3 $R/gpurun_out/attn_pmc_p$i.log
done
cd $R
python benchmarks/pmc_by_kernel.py attn_ gpurun_out/attn_pmc/p* > gpurun_out/attn_pmc_summary.txt
f=$(find gpurun_out/attn_pmc/p1 -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_by_grid.py $f 5 | grep attn >> gpurun_out/attn_pmc_summary.txt
rm -rf gpurun_out/attn_pmc
cat gpurun_out/attn_pmc_summary.txt
