cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_dense.py -q -m gpu -k "first_use" 2>&1 | grep -v Warning | tail -40
