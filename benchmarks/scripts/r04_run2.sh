#!/bin/bash
# round 4, run 2: re-tune of the NN / TN shapes with the hand-scheduled kernels among the candidates, Stage-II bench old vs new table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_2
ACT_GEMM_TUNE_TABLE=0 ACT_TUNE_ONLY=nntn timeout 1500 python benchmarks/tune_table.py act_amd/gemm_tune_gfx950.json gpurun_out/r04_2/gemm_tune_gfx950.json > gpurun_out/r04_2/tune.log 2>&1
tail -3 gpurun_out/r04_2/tune.log
python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r04_2/bench_old.json 2> gpurun_out/r04_2/bench_old.err; cut -c1-300 gpurun_out/r04_2/bench_old.json
ACT_GEMM_TUNE_FILE=gpurun_out/r04_2/gemm_tune_gfx950.json python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r04_2/bench_new.json 2> gpurun_out/r04_2/bench_new.err; cut -c1-300 gpurun_out/r04_2/bench_new.json
