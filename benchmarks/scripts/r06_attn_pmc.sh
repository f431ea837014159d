#!/bin/bash
# round-6: MfmaUtil + wait breakdown of the attention kernels on the three workload shapes (own rocprofv3 --pmc passes; --kernel-trace only)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/r06_attn_pmc; mkdir -p $O
for SH in "128 64 64 12 64" "32 0 512 12 64" "32 0 104 12 64" "128 0 64 6 64"; do
  T=$(echo $SH | tr ' ' '_')
  i=0
  for P in "MfmaUtil" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p_${T}_$i -- python $R/benchmarks/attn_pmc_probe.py $SH > $O/log_${T}_$i.txt 2>&1 || tail -3 $O/log_${T}_$i.txt
  done
  echo "==== shape B S0 Sq H hd = $SH" >> $O/summary.txt
  python $R/benchmarks/pmc_by_kernel.py attn_ $O/p_${T}_* >> $O/summary.txt
  f=$(find $O/p_${T}_1 -name "*kernel_trace.csv" | head -1); python $R/benchmarks/trace_by_grid.py $f 5 | grep attn >> $O/summary.txt
  rm -rf $O/p_${T}_*
done
cat $O/summary.txt
