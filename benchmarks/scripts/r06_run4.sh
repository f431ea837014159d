#!/bin/bash
# round-6 call 4: Stage-I deferred weight gradients (bit-identity test + A/B), finetune dW overlap A/B, trajectory diagnostics, headline after the attention change
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run4; O=gpurun_out/r06_run4
python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_runner.py -q -s -k "trajectory or deferred or stage1" 2>&1 | grep -v "Warning\|warnings.warn\|^$\|^E   " | tail -40 > $O/pytest.log; cat $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument"
for v in 1 0 1 0; do echo "stage1 ACT_DEFER_DW=$v $(ACT_DEFER_DW=$v $B --stage 1 --steps 15 --warmup 4 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
for v in 1 0 1 0; do echo "stage3 ACT_FT_OVERLAP_DW=$v $(ACT_FT_OVERLAP_DW=$v $B --stage 3 --steps 30 --warmup 8 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
for v in 1 0; do echo "stage2 ACT_ATTN_BWD_ONE=$v $(ACT_ATTN_BWD_ONE=$v $B --steps 30 --warmup 8 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
for v in 1 0; do echo "c5 ACT_ATTN_BWD_ONE=$v $(ACT_ATTN_BWD_ONE=$v $B --config c5 --steps 6 --warmup 2 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
