#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_run17; mkdir -p $O
python -m pytest tests/test_gpu_dense.py -q -x -s -k "dgcnn or edge" 2>&1 | grep "edge bwd lds\|passed\|failed\|Error\|assert" | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/benchmarks/edge_bwd_bench.py > $O/log.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $O/stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} min_us {float(r['MinNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf $O/prof; grep "C=" $O/log.txt
cd $R
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument --stage 1 --steps 15 --warmup 4"
for v in 1 0 1 0; do echo "stage1 ACT_EDGE_BWD_LDS=$v $(ACT_EDGE_BWD_LDS=$v $B | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
python -m pytest tests/test_gpu_model.py -q -x -k "stage1" 2>&1 | tail -2 | tee -a $O/pytest.log
