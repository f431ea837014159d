#!/bin/bash
# evidence for the OPT-IN split-bf16 teacher configuration: bench line + rocprofv3 kernel stats (streams serialised) + MfmaUtil pass
R=$GRAFT_REPO_ROOT; cd $R
export ACT_TEACHER_BF16X3=1
timeout 600 python bench.py --steps 30 --warmup 8 --no-other-workloads > gpurun_out/r05_bench_c2_x3optin.json 2> gpurun_out/r05_bench_c2_x3optin.err
cd /tmp; export TMPDIR=/tmp
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x3 -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/prof_x3.log 2>&1
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $R/gpurun_out/pmc_mfma_x3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_mfma_x3.log 2>&1
cd $R
python benchmarks/pmc_mfma_util.py gpurun_out/pmc_mfma_x3 > gpurun_out/r05_pmc_mfma_util_c2_x3optin.json
find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_c2_x3optin_kernel_stats.csv
f=$(find gpurun_out/prof_x3 -name "*kernel_trace.csv" | head -1); python benchmarks/trace_by_grid.py $f 23 > gpurun_out/r05_c2_x3optin_trace_by_grid.txt
rm -rf gpurun_out/prof_x3 gpurun_out/pmc_mfma_x3
head -14 gpurun_out/r05_c2_x3optin_trace_by_grid.txt
