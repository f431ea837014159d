cd $GRAFT_REPO_ROOT
for d in 0 1 2; do echo "== ACT_ATTN_DBG=$d"; ACT_ATTN_DBG=$d python benchmarks/attn_bench.py 2>&1 | grep -E "stage1|S=128|S=512"; done
