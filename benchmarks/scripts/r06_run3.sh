#!/bin/bash
# round-6 call 3: attention forward variants (one pair per 128-thread workgroup, s_setprio), backward prio; trajectory test again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run3; O=gpurun_out/r06_run3
python -m pytest tests/test_gpu_trajectory.py -q -x -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -12 > $O/pytest_traj.log; cat $O/pytest_traj.log
for v in "base" "ACT_ATTN_FWD_NW=2" "ACT_ATTN_FWD_PRIO=1" "ACT_ATTN_FWD_NW=2 ACT_ATTN_FWD_PRIO=1" "ACT_ATTN_BWD_PRIO=1"; do
  echo "== $v" | tee -a $O/attn_variants.txt
  if [ "$v" = base ]; then python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_variants.txt
  else env $v python benchmarks/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_variants.txt; fi
done
ACT_ATTN_FWD_NW=2 ACT_ATTN_FWD_PRIO=1 python -m pytest tests/test_gpu_dense.py -q -x -k "attention" 2>&1 | tail -3 | tee $O/pytest_variants.log
