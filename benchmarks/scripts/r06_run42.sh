#!/bin/bash
# round-6 call 42: tile rasterisation by reciprocal multiplication (scalar ALU) instead of run-time integer divisions (vector ALU): VALU instructions per wave of the
# teacher's fc1 launch (64 x 64 asm tile, tile id 32) with ACT_GEMM_FASTDIV=0 / 1, timings of a few shapes, then the GEMM tests
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_run42; O=$R/gpurun_out/r06_run42
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE MfmaUtil"; do
    ACT_GEMM_FASTDIV=$v timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$v/p_$(echo $P | cut -c1-6) -- python $R/benchmarks/gemm_pmc_probe.py 8192 3072 768 32 > $O/pmc_$v.log 2>&1 || tail -3 $O/pmc_$v.log
  done
  cd $R; echo "== ACT_GEMM_FASTDIV=$v" | tee -a $O/pmc.txt; python benchmarks/pmc_by_kernel.py "sgemm_nt_asm" $O/pmc_$v/p_* | tee -a $O/pmc.txt; rm -rf $O/pmc_$v; cd /tmp
done
cd $R
for rep in 1 2; do for v in 0 1; do
  echo "== ACT_GEMM_FASTDIV=$v" | tee -a $O/time.txt
  ACT_GEMM_FASTDIV=$v python benchmarks/gelu_cost_bench.py 2>&1 | grep fc1 | tee -a $O/time.txt
done; done
python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_dense.txt
