#!/bin/bash
# round 4: sparse max-pool-backward products (csrc/pool_bwd.hip): tests, then A/B of the Stage-II / C5 step against the dense on-load kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -x -q -m gpu -k "maxpool or encoder" > gpurun_out/r04_sparse_tests.log 2>&1
tail -5 gpurun_out/r04_sparse_tests.log
for v in 1 0 1 0; do
  ACT_PN_POOL_BWD_SPARSE=$v timeout 600 python bench.py --steps 40 --warmup 10 --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sparse=$v stage2', d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/r04_sparse_ab.txt
for v in 1 0; do
  ACT_PN_POOL_BWD_SPARSE=$v timeout 600 python bench.py --config c5 --steps 6 --warmup 2 --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sparse=$v c5', d['ms_per_step'], d['value'])"
  ACT_PN_POOL_BWD_SPARSE=$v timeout 600 python bench.py --stage 1 --steps 10 --warmup 3 --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sparse=$v stage1', d['ms_per_step'], d['value'])"
done 2>&1 | tee -a gpurun_out/r04_sparse_ab.txt
