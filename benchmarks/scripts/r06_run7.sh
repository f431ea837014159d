#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run7; O=gpurun_out/r06_run7
python -m pytest tests/test_gpu_dense.py -q -x -k "attention" 2>&1 | tail -5 | tee $O/pytest.log
echo "== ACT_ATTN_FWD_PAIR=1" | tee $O/attn.txt; python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn.txt
echo "== ACT_ATTN_FWD_PAIR=0" | tee -a $O/attn.txt; ACT_ATTN_FWD_PAIR=0 python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn.txt
python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_composite.py tests/test_gpu_model.py -q -x 2>&1 | tail -5 | tee -a $O/pytest.log
