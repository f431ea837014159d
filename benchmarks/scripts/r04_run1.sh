#!/bin/bash
# round 4, run 1: parity of the hand-scheduled NT kernels, re-tune of the NT shapes, Stage-II bench with the new table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_1
python -m pytest tests/test_gpu_dense.py -x -q -m gpu -k "hand_scheduled or scalar_fallback or 32_deep" 2>&1 | tail -15 > gpurun_out/r04_1/pytest.log
cat gpurun_out/r04_1/pytest.log
ACT_GEMM_TUNE_TABLE=0 ACT_TUNE_ONLY=nt timeout 900 python benchmarks/tune_table.py act_amd/gemm_tune_gfx950.json gpurun_out/r04_1/gemm_tune_gfx950.json > gpurun_out/r04_1/tune.log 2>&1
tail -5 gpurun_out/r04_1/tune.log
python bench.py --no-cpu-baseline > gpurun_out/r04_1/bench_old.json 2> gpurun_out/r04_1/bench_old.err; cat gpurun_out/r04_1/bench_old.json | cut -c1-400
ACT_GEMM_TUNE_FILE=gpurun_out/r04_1/gemm_tune_gfx950.json python bench.py --no-cpu-baseline > gpurun_out/r04_1/bench_new.json 2> gpurun_out/r04_1/bench_new.err; cat gpurun_out/r04_1/bench_new.json | cut -c1-400
