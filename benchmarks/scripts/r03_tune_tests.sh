# pass 1: the whole GPU suite with the first-use autotuner on, dumping shipped table + every newly tuned shape
cd $GRAFT_REPO_ROOT
ACT_GEMM_TUNE_SAVE=$GRAFT_REPO_ROOT/gpurun_out/gemm_tune_with_tests.json timeout 3000 python -m pytest tests/ -q -m gpu > gpurun_out/r03_tune_tests.log 2>&1
tail -5 gpurun_out/r03_tune_tests.log
python - <<'PY'
import json
a = json.load(open("act_amd/gemm_tune_gfx950.json"))["configs"]; b = json.load(open("gpurun_out/gemm_tune_with_tests.json"))["configs"]
print("shipped", len(a), "-> with test shapes", len(b), "changed:", sum(1 for k in a if b.get(k) != a[k]))
PY
