cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -x -q 2>&1 | tail -6
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r03_b5_c2.json 2> gpurun_out/r03_b5_c2.err
ACT_GROUPED_DW=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r03_b5_c2_nogroup.json 2> gpurun_out/r03_b5_c2_nogroup.err
timeout 600 python bench.py --stage 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_b5_s1.json 2> gpurun_out/r03_b5_s1.err
python - <<'PY'
import json
for f in ("r03_b5_c2", "r03_b5_c2_nogroup", "r03_b5_s1"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        k = d["kernels"]
        print(f, round(d["value"], 1), round(d["ms_per_step"], 3), "hip_ms", round(d["hip_kernel_ms_per_step"], 2), {n: round(k[n]["ms_per_step"], 3) for n in ("sgemm_nt", "sgemm_nn", "sgemm_tn", "colsum", "attention_fwd", "attention_bwd") if n in k})
    except Exception as e:
        print(f, "failed", e)
PY
