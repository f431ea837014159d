#!/bin/bash
# round-4 evidence: driver command, sustained run, per-workload profiles (kernel stats, PMC traffic, MfmaUtil), fc1 PMC head-to-head with hipBLASLt
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r04_bench_driver_cmd.json 2> gpurun_out/r04_bench_driver_cmd.err
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-instrument --no-other-workloads > gpurun_out/r04_bench_sustained.json 2>/dev/null
bash benchmarks/scripts/r04_profiles.sh c2 > gpurun_out/r04_prof_c2.log 2>&1
bash benchmarks/scripts/r04_profiles.sh s1 --stage 1 > gpurun_out/r04_prof_s1.log 2>&1
bash benchmarks/scripts/r04_profiles.sh c5 --config c5 > gpurun_out/r04_prof_c5.log 2>&1
GEMM_ARGS="8192 3072 768 32" bash benchmarks/scripts/r04_gemm_pmc.sh > gpurun_out/r04_gemm_pmc.log 2>&1
cp gpurun_out/gemm_pmc_summary.txt gpurun_out/r04_gemm_pmc_fc1_vs_hipblaslt.txt
ls gpurun_out | grep r04_ | head -50
