#!/bin/bash
# round-6 call 36: cost of the GELU epilogue on the fc1 launches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run36; O=gpurun_out/r06_run36
python benchmarks/gelu_cost_bench.py 2>&1 | grep -v Warning | tee $O/gelu_cost.txt
