#!/bin/bash
# round-6 call 31: ablation of the teacher-shape attention forward (what is the shared resource? the start-up stagger of call 30 changed nothing)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run31; O=gpurun_out/r06_run31
# the ablation variants exist only in a dev build of the library; the product build is restored at the end
ACT_HIPCC_EXTRA=-DACT_ATTN_DIAG python -c "import act_amd.build as b; b.build()"
for d in ${DIAGS:-0 1 2 3 4 7 8 16 24 28 31 0}; do ACT_ATTN_FWD_DIAG=$d python benchmarks/attn_fwd_diag.py 2>&1 | grep DIAG | tee -a $O/diag.txt; done
python -c "import act_amd.build as b; b.build()"
