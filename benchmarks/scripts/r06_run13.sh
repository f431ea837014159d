#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run13; O=gpurun_out/r06_run13
python -m pytest tests/test_gpu_dense.py -q -x -s -k "dgcnn or edge" 2>&1 | grep "edge bwd lds\|passed\|failed\|Error\|assert" | tee $O/pytest.log
python benchmarks/edge_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/edge_bwd.txt
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument --stage 1 --steps 15 --warmup 4"
for v in 1 0 1 0; do echo "stage1 ACT_EDGE_BWD_LDS=$v $(ACT_EDGE_BWD_LDS=$v $B | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
