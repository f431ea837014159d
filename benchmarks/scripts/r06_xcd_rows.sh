#!/bin/bash
# round-6 review item 3: does a 2-D XCD tile placement (fewer fabric re-reads of the A band) pay?  step time and PMC traffic of the GEMM families per ACT_GEMM_XCD_ROWS setting
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_xcd; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-workloads --no-instrument --steps 30 --warmup 8"
for r in 0 2 4 1 0 2; do echo "stage2 ACT_GEMM_XCD_ROWS=$r $(ACT_GEMM_XCD_ROWS=$r $B | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])')" | tee -a $O/ab.txt; done
cd /tmp; export TMPDIR=/tmp
for r in 0 2 4; do
  ACT_GEMM_XCD_ROWS=$r timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $O/f$r.log 2>&1
  ACT_GEMM_XCD_ROWS=$r timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w$r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $O/w$r.log 2>&1
  python $R/benchmarks/pmc_traffic.py $O/f$r $O/w$r c2 > $O/traffic_xcd_rows_$r.json
  python - <<PY >> $O/ab.txt
import json
t=json.load(open("$O/traffic_xcd_rows_$r.json"))["kernels"]
print("ACT_GEMM_XCD_ROWS=$r HBM/fabric MB per launch:", {k: round(v["hbm_bytes_per_launch"]/1e6,1) for k,v in t.items() if k in ("sgemm_nt","sgemm_nn","sgemm_tn")})
PY
  rm -rf $O/f$r $O/w$r
done
cat $O/ab.txt
