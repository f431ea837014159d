#!/bin/bash
# same-box A/B of the transposed-V staging of the attention forward (ACT_ATTN_VT=1, default) against V staged as [key][d] (ACT_ATTN_VT=0)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo "== VT=1 (run $r)"; ACT_ATTN_VT=1 python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids
echo "== VT=0 (run $r)"; ACT_ATTN_VT=0 python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
python -m pytest tests/test_gpu_dense.py -q -m gpu -k "attention" 2>&1 | tail -2
for i in 1 2; do
  for vt in 1 0; do
  ACT_ATTN_VT=$vt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('VT=$vt c2', d['ms_per_step'])"
  done
done
for vt in 1 0; do
ACT_ATTN_VT=$vt python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('VT=$vt c5', d['ms_per_step'])"
done
