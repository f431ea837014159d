#!/bin/bash
# full GPU suite + Stage-II bench (A/B: fused mini-PointNet launches on the compiler loop vs the hand-scheduled loop)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_suite
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r04_suite/pytest.log; cat gpurun_out/r04_suite/pytest.log
ACT_GEMM_FX_ASM=0 python bench.py --no-cpu-baseline > gpurun_out/r04_suite/bench_fx0.json 2> gpurun_out/r04_suite/bench_fx0.err; cut -c1-300 gpurun_out/r04_suite/bench_fx0.json
python bench.py --no-cpu-baseline > gpurun_out/r04_suite/bench_fx1.json 2> gpurun_out/r04_suite/bench_fx1.err; cut -c1-300 gpurun_out/r04_suite/bench_fx1.json
