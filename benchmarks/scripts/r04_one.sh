cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_dense.py -q -m gpu -x -k "hand_scheduled or scalar_fallback or first_use or autotuner" 2>&1 | grep -v Warning | tail -30
