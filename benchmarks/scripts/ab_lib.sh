#!/bin/bash
# same-box A/B of two builds of libact_hip.so (ACT_LIB_PATH): isolated NT rates + the Stage-II step, interleaved
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/benchmarks/diag/libact_hip_head.so
echo "== new"; python benchmarks/epi_spec_bench.py 2>&1 | grep -v amdgpu.ids | head -7
echo "== head"; ACT_LIB_PATH=$OLD python benchmarks/epi_spec_bench.py 2>&1 | grep -v amdgpu.ids | head -7
for i in 1 2 3; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['ms_per_step'])"
  ACT_LIB_PATH=$OLD python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('head', d['ms_per_step'])"
done
