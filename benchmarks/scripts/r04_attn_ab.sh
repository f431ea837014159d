#!/bin/bash
# same-box A/B: attention forward with software-pipelined K/V staging (new) vs the round-3 kernel (head)
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/benchmarks/diag/libact_hip_head.so
echo "== new"; python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids
echo "== head"; ACT_LIB_PATH=$OLD python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_dense.py -q -m gpu -k "attention" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new  c2', d['ms_per_step'])"
  ACT_LIB_PATH=$OLD python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('head c2', d['ms_per_step'])"
done
python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new  c5', d['ms_per_step'])"
ACT_LIB_PATH=$OLD python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-instrument --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('head c5', d['ms_per_step'])"
