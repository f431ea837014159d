#!/bin/bash
# round-6 call 32: static wave priority by CU slot in the attention forward (ACT_ATTN_FWD_SLOT_PRIO), alone and with the start-up stagger
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run32; O=gpurun_out/r06_run32
for cfg in "0 0" "1 0" "2 0" "1 3" "1 6" "2 6" "0 0"; do
  set -- $cfg
  echo "== SLOT_PRIO=$1 STAGGER=$2" | tee -a $O/ab.txt
  ACT_ATTN_FWD_SLOT_PRIO=$1 ACT_ATTN_FWD_STAGGER=$2 python benchmarks/attn_bench.py 2>&1 | grep -v Warning | grep "prompt-prefix\|S=128\|dec S=64\|S=512\|64+512\|S=104\|S=65" | tee -a $O/ab.txt
done
