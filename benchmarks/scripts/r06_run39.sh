#!/bin/bash
# round-6 call 39: where do the tiny Stage-I gradients move under the one-polynomial GELU?  (both builds)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run39; O=gpurun_out/r06_run39
for v in 0 1; do
  ACT_HIPCC_EXTRA=-DACT_GELU_FAST=$v python -c "import act_amd.build as b; b.build()"
  echo "== build -DACT_GELU_FAST=$v" | tee -a $O/diff.txt
  DIAG_SEEDS=${DIAG_SEEDS:-777} python benchmarks/diag/stage1_tiny_grad_diff.py 2>&1 | grep "seed\|rel " | tail -60 | tee -a $O/diff.txt
done
python -c "import act_amd.build as b; b.build()"
