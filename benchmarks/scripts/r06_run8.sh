#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run8; O=gpurun_out/r06_run8
python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -q -x -k "attention or prefix or teacher or stack" 2>&1 | tail -3 | tee $O/pytest.log
python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/attn.txt
