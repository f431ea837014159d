#!/bin/bash
# round-6 call 5: trajectory test (self-calibrated running-stat bar), Stage-I deferred dW with stream priorities, the driver's command with the latency models
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run5; O=gpurun_out/r06_run5
python -m pytest tests/test_gpu_trajectory.py -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -60 > $O/pytest.log; grep "trajectory\]\|passed\|failed\|Error" $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument"
J='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])'
for e in "ACT_DEFER_DW=0" "ACT_DEFER_DW=1 ACT_MAIN_PRIO=-1" "ACT_DEFER_DW=0 ACT_MAIN_PRIO=-1" "ACT_DEFER_DW=1 ACT_MAIN_PRIO=-1"; do echo "stage1 $e $(env $e $B --stage 1 --steps 15 --warmup 4 | python -c "$J")" | tee -a $O/ab.txt; done
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_run5/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d["group_fps_knn"], indent=1))
for k,v in d["kernels"].items():
    if "alg_GBs" in v: print(k, round(v["alg_GBs"]), round(v["ms_per_step"],3))
print({k:(v.get("ms_per_step"),) for k,v in d["other_workloads"].items() if isinstance(v,dict) and "ms_per_step" in v})
PY
