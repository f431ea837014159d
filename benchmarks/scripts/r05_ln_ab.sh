#!/bin/bash
# same-box A/B: LayerNorm-backward rows per workgroup (16 = shipped, 4 / 8 = one / two rows per wave)
cd $GRAFT_REPO_ROOT
for i in 1 2; do for r in 16 4 8; do
ACT_LN_BWD_RPB=$r python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('RPB=$r c2', round(d['ms_per_step'],3), 'ln_bwd', round(d['kernels']['layernorm_bwd']['ms_per_step'],3), 'loss', d['config']['final_loss'])"
done; done
