#!/bin/bash
# LayerNorm forward / backward with unconditional batched loads (new lib) vs HEAD~ (benchmarks/diag/libact_hip_head.so): tests + interleaved step A/B of the three workloads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run23; O=gpurun_out/r06_run23; OLD=$GRAFT_REPO_ROOT/benchmarks/diag/libact_hip_head.so
python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -q -x -k "layernorm or layer_norm or block or stack or teacher or prefix" 2>&1 | tail -2 | tee $O/pytest.log
J='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]; print(round(d["ms_per_step"],3), "ms/step; layernorm fwd/bwd ms/step", round(k["layernorm_fwd"]["ms_per_step"],3), round(k["layernorm_bwd"]["ms_per_step"],3), "GB/s", round(k["layernorm_fwd"]["alg_GBs"]), round(k["layernorm_bwd"]["alg_GBs"]), "loss", d["config"]["final_loss"])'
for w in "" "--stage 1" "--config c5"; do
  S=30; [ "$w" != "" ] && S=10; [ "$w" = "--config c5" ] && S=6
  for i in 1 2; do
    echo "new  [$w] $(python bench.py $w --steps $S --warmup 4 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "$J")" | tee -a $O/ab.txt
    echo "head [$w] $(ACT_LIB_PATH=$OLD python bench.py $w --steps $S --warmup 4 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "$J")" | tee -a $O/ab.txt
  done
done
