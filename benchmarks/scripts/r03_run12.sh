#!/bin/bash
# kernel trace of the production (overlapped) timed loop -> idle-gap analysis
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gaps -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_gaps.log 2>&1
cd $R; f=$(find gpurun_out/prof_gaps -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_gaps.py $f 2800 > gpurun_out/r03_c2_gaps.txt; cat gpurun_out/r03_c2_gaps.txt
cp $f gpurun_out/r03_c2_overlap_trace.csv; rm -rf gpurun_out/prof_gaps
