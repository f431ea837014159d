#!/bin/bash
# round-6 opening call: GPU suite with the first-use GEMM picks dumped (merged into the shipped table afterwards), attention kernel baselines, the driver's command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run1
ACT_GEMM_TUNE_SAVE=gpurun_out/r06_run1/tuned_%p.json python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|warnings.warn\|pin_memory\|^$" | tail -25 > gpurun_out/r06_run1/pytest.log; cat gpurun_out/r06_run1/pytest.log
python benchmarks/attn_bench.py > gpurun_out/r06_run1/attn_bench.txt 2>&1; cat gpurun_out/r06_run1/attn_bench.txt
ACT_GEMM_TUNE_SAVE=gpurun_out/r06_run1/tuned_bench_%p.json python bench.py > gpurun_out/r06_run1/bench.json 2> gpurun_out/r06_run1/bench.err; tail -c 600 gpurun_out/r06_run1/bench.json
