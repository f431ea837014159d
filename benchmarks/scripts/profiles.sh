# per-round evidence (RND=rNN, default r05): bench lines of every workload + rocprofv3 kernel stats, per-workload PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
# MfmaUtil.  usage: [RND=r05] profiles.sh <tag> <bench args...>   e.g.  profiles.sh c2 ; profiles.sh s1 --stage 1 ; profiles.sh c5 --config c5
set -x
RND=${RND:-r05}
R=$GRAFT_REPO_ROOT; TAG=$1; shift; ARGS="$@"
cd $R
timeout 900 python bench.py $ARGS --steps 30 --warmup 8 --no-other-workloads > gpurun_out/${RND}_bench_$TAG.json 2> gpurun_out/${RND}_bench_$TAG.err
cd /tmp; export TMPDIR=/tmp
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py $ARGS --steps 16 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/prof_$TAG.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_write_$TAG.log 2>&1
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $R/gpurun_out/pmc_mfma_$TAG -- python $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/pmc_mfma_$TAG.log 2>&1
cd $R
python benchmarks/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG $TAG > gpurun_out/${RND}_pmc_traffic_$TAG.json
python benchmarks/pmc_mfma_util.py gpurun_out/pmc_mfma_$TAG > gpurun_out/${RND}_pmc_mfma_util_$TAG.json
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${RND}_${TAG}_kernel_stats.csv
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); python benchmarks/trace_by_grid.py $f 23 > gpurun_out/${RND}_${TAG}_trace_by_grid.txt
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_mfma_$TAG
ls -la gpurun_out | grep ${RND}_ | tail -12
