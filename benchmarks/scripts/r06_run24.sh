#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run24; O=gpurun_out/r06_run24; OLD=$GRAFT_REPO_ROOT/benchmarks/diag/libact_hip_head.so
python -m pytest tests/test_gpu_dense.py -q -x -k "attention" 2>&1 | tail -2 | tee $O/pytest.log
for i in 1 2; do
echo "== new" | tee -a $O/attn.txt; python benchmarks/attn_bench.py 2>&1 | grep -v "amdgpu.ids\|S=14\|S=16\|S=7 " | tee -a $O/attn.txt
echo "== head" | tee -a $O/attn.txt; ACT_LIB_PATH=$OLD python benchmarks/attn_bench.py 2>&1 | grep -v "amdgpu.ids\|S=14\|S=16\|S=7 " | tee -a $O/attn.txt
done
