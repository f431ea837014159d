cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_point_ops.py -x -q 2>&1 | tail -4
python benchmarks/fps_bench.py 2>&1 | grep -v amdgpu | tail -20
