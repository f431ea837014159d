#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run22; O=gpurun_out/r06_run22
python -m pytest tests/test_gpu_dense.py -q -x -k "dgcnn or edge" 2>&1 | tail -2 | tee $O/pytest.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_composite.py -q -x -k "stress or teacher or dgcnn" 2>&1 | tail -2 | tee -a $O/pytest.log
B="python bench.py --no-cpu-baseline --no-other-workloads --config c5 --steps 6 --warmup 2"
for v in 1 0 1 0; do ACT_EDGE_GN_SLAB=$v $B > $O/c5_$v.json 2>/dev/null; python - $O/c5_$v.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']['gn_lrelu_max']
print(f"c5 ACT_EDGE_GN_SLAB={sys.argv[2]}: {d['ms_per_step']:.2f} ms/step {d['value']:.1f} clouds/s loss {d['config']['final_loss']}; gn_lrelu_max {k['ms_per_step']:.3f} ms/step ({k['launches_per_step']:.0f} launches, {k['alg_GBs']:.0f} GB/s compulsory)")
PY
done
