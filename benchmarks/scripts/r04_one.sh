cd $GRAFT_REPO_ROOT; python -m pytest tests -q -m gpu -x -k "grouped or maxpool or knn or group or block or composite or point" 2>&1 | grep -v Warning | tail -12
ACT_GEMM_GROUPED_ASM=0 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | cut -c1-250
python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | cut -c1-250
