#!/bin/bash
# full GPU suite with the first-use GEMM picks dumped (merged into the shipped table afterwards) + smoke + the JT A/B of the attention forward
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_suite; O=gpurun_out/r06_suite
ACT_GEMM_TUNE_SAVE=$O/tuned_%p.json python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|warnings.warn\|pin_memory\|^$" | tail -15 > $O/pytest.log; cat $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
echo "== ACT_ATTN_JT=2" > $O/attn_jt2.txt; ACT_ATTN_JT=2 python benchmarks/attn_bench.py 2>&1 | grep "prompt-prefix\|S=128 \|dec S=64" | tee -a $O/attn_jt2.txt
echo "== default" >> $O/attn_jt2.txt; python benchmarks/attn_bench.py 2>&1 | grep "prompt-prefix\|S=128 \|dec S=64" | tee -a $O/attn_jt2.txt
