#!/bin/bash
# round-6 call 30: attention forward start-up de-phasing (ACT_ATTN_FWD_STAGGER: the workgroup in CU slot t sleeps (t % mod) * n * 1024 cycles before its first load)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run30; O=gpurun_out/r06_run30
for cfg in "0 3" "2 3" "3 3" "4 3" "6 3" "8 3" "-3 3" "-6 3" "3 2" "6 2" "0 3"; do
  set -- $cfg
  echo "== ACT_ATTN_FWD_STAGGER=$1 MOD=$2" | tee -a $O/ab.txt
  ACT_ATTN_FWD_STAGGER=$1 ACT_ATTN_FWD_STAGGER_MOD=$2 python benchmarks/attn_bench.py 2>&1 | grep -v Warning | grep "prompt-prefix\|S=128\|dec S=64\|S=512\|64+512\|S=104\|S=65" | tee -a $O/ab.txt
done
