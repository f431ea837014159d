#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (separate rocprofv3 runs), per workload.  usage: r04_traffic.sh <tag> <bench args...>
R=$GRAFT_REPO_ROOT; TAG=$1; shift; ARGS="$@"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python benchmarks/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG $TAG > gpurun_out/r04_pmc_traffic_$TAG.json
rm -rf gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
python -c "
import json; d=json.load(open('gpurun_out/r04_pmc_traffic_$TAG.json')); print('$TAG', d['calibration']['fetch_factor'], d['calibration']['write_factor']); k=d['kernels']
for n in ('sgemm_nt','sgemm_nn','sgemm_tn','pool_bwd_dx4_kernel','pool_bwd_dw2_kernel','colstats_stage1'):
    print(n, {a: round(b/1e6,1) for a,b in k.get(n,{}).items() if a.endswith('launch')})"
