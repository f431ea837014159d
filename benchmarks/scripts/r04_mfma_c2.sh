#!/bin/bash
# MfmaUtil pass of the Stage-II step alone (per folded kernel id and per GEMM kernel function)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for TAG_ARGS in "c2:" "s1:--stage 1" "c5:--config c5"; do
  TAG=${TAG_ARGS%%:*}; ARGS=${TAG_ARGS#*:}
  ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $R/gpurun_out/pmc_mfma_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_mfma_$TAG.log 2>&1
  python $R/benchmarks/pmc_mfma_util.py $R/gpurun_out/pmc_mfma_$TAG > $R/gpurun_out/r04_pmc_mfma_util_$TAG.json
  rm -rf $R/gpurun_out/pmc_mfma_$TAG
done
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
for t in ("c2", "s1", "c5"):
    d = json.load(open(f"{R}/gpurun_out/r04_pmc_mfma_util_{t}.json"))
    print(t, {k: round(v["mfma_util_percent_time_weighted"], 1) for k, v in d["gemm_kernel_functions"].items()})
PY
