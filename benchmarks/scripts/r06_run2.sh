#!/bin/bash
# round-6 call 2: single-pass attention backward (numerics + A/B timing), 20-step trajectory parity, preflight, chunked stacks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run2; O=gpurun_out/r06_run2
python -m pytest tests/test_gpu_dense.py tests/test_gpu_trajectory.py tests/test_gpu_composite.py -q -x -k "attention or trajectory or stack or prefix" -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -30 > $O/pytest_a.log; cat $O/pytest_a.log
python -m pytest tests/test_gpu_ddp.py -q -x -k "preflight" -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -15 > $O/pytest_b.log; cat $O/pytest_b.log
ACT_ATTN_BWD_ONE=1 python benchmarks/attn_bench.py > $O/attn_one.txt 2>&1; cat $O/attn_one.txt
ACT_ATTN_BWD_ONE=0 python benchmarks/attn_bench.py > $O/attn_reg.txt 2>&1; grep -v amdgpu.ids $O/attn_reg.txt
