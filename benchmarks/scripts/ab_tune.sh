#!/bin/bash
# in-step A/B of alternative tune tables (ACT_GEMM_TUNE_FILE), interleaved with the shipped one
cd $GRAFT_REPO_ROOT
run() { ACT_GEMM_TUNE_FILE=$1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$2', d['ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('shipped', d['ms_per_step'])"
  for f in "$@"; do run benchmarks/diag/$f.json $f; done
done
