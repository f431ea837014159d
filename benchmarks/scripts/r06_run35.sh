#!/bin/bash
# round-6 call 35: lean online softmax of the attention forward (packed FMA + exp2 instead of sub / mul / mul / exp), A/B by two builds on one box + the numerics tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run35; O=gpurun_out/r06_run35
F="prompt-prefix\|S=128\|dec S=64\|S=512\|64+512\|S=104\|S=65"
for rep in 1 2; do
  ACT_HIPCC_EXTRA=-DACT_ATTN_SOFTMAX_LEAN=0 python -c "import act_amd.build as b; b.build()"
  echo "== previous softmax (build -DACT_ATTN_SOFTMAX_LEAN=0)" | tee -a $O/ab.txt
  python benchmarks/attn_bench.py 2>&1 | grep -v Warning | grep "$F" | tee -a $O/ab.txt
  python -c "import act_amd.build as b; b.build()"
  echo "== lean softmax (product build)" | tee -a $O/ab.txt
  python benchmarks/attn_bench.py 2>&1 | grep -v Warning | grep "$F" | tee -a $O/ab.txt
done
python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_dense.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_model.txt
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument"
for c in "" "--config c5 --steps 6 --warmup 2" "--stage 1 --steps 10 --warmup 3"; do echo "bench $c: $($B $c | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/bench.txt; done
