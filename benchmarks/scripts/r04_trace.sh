#!/bin/bash
# kernel trace of the Stage-II step (streams serialised), per-grid table.  usage: r04_trace.sh <tag> [env assignments...]
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
env "$@" ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_${TAG}_kernel_stats.csv
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); python benchmarks/trace_by_grid.py $f 23 > gpurun_out/r04_${TAG}_trace_by_grid.txt
rm -rf gpurun_out/prof_$TAG
head -40 gpurun_out/r04_${TAG}_trace_by_grid.txt
