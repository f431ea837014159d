#!/bin/bash
# round-6 call 41: buffer loads (scalar base + lane offset + scalar K offset) in the K loop of the compiler-scheduled NT kernels (mini-PointNet fused launches), A/B by two builds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run41; O=gpurun_out/r06_run41
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument"
line() { python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])'; }
for rep in 1 2; do
  for v in 0 1; do
    ACT_HIPCC_EXTRA=-DACT_NT16_BUFFER_LOADS=$v python -c "import act_amd.build as b; b.build()"
    echo "== build -DACT_NT16_BUFFER_LOADS=$v" | tee -a $O/ab.txt
    python benchmarks/fx_bench.py 2>&1 | grep "conv" | tee -a $O/ab.txt
    echo "stage2 $($B --steps 30 --warmup 8 | line)" | tee -a $O/ab.txt
    echo "stage1 $($B --stage 1 --steps 12 --warmup 4 | line)" | tee -a $O/ab.txt
    echo "c5     $($B --config c5 --steps 6 --warmup 2 | line)" | tee -a $O/ab.txt
  done
done
python -c "import act_amd.build as b; b.build()"
python -m pytest tests/test_gpu_dense.py tests/test_gpu_composite.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_dense.txt
python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_model.txt
