cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_dense.py -x -q -k "attention or prefix or block" 2>&1 | tail -4
python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids
