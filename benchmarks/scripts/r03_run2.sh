set -x
R=$GRAFT_REPO_ROOT; cd $R
bash benchmarks/scripts/r03_host_8procs.sh > gpurun_out/r03_host_8procs.log 2>&1
timeout 1200 python benchmarks/student_gemm_sweep.py > gpurun_out/r03_student_sweep.log 2>&1
cd /tmp; export TMPDIR=/tmp
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_grid -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_grid.log 2>&1
cd $R
f=$(find gpurun_out/prof_grid -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_by_grid.py $f 13 > gpurun_out/r03_trace_by_grid_before.txt
rm -rf gpurun_out/prof_grid
tail -30 gpurun_out/r03_student_sweep.log
