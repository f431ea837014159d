#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run11; O=gpurun_out/r06_run11
ACT_ATTN_BWD_NP=1 python -m pytest tests/test_gpu_dense.py -q -x -k "attention" 2>&1 | tail -2 | tee $O/pytest.log
python -m pytest tests/test_gpu_dense.py tests/test_gpu_ddp.py -q -x -k "attention or preflight" 2>&1 | tail -2 | tee -a $O/pytest.log
for v in 2 1 2 1; do echo "== ACT_ATTN_BWD_NP=$v" | tee -a $O/attn.txt; ACT_ATTN_BWD_NP=$v python benchmarks/attn_bench.py 2>&1 | grep "dec S=64\|finetune\|S=7 \|prompt-prefix" | tee -a $O/attn.txt; done
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument --steps 30 --warmup 8"
for v in 2 1 2 1; do echo "stage2 ACT_ATTN_BWD_NP=$v $(ACT_ATTN_BWD_NP=$v $B | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
