cd $GRAFT_REPO_ROOT
for dp in 0 1 2 3 4; do echo "== dephase $dp"; ACT_ATTN_DEPHASE=$dp python benchmarks/attn_bench.py 2>&1 | grep "prompt-prefix\|S=128 \|dec S=64"; done
