set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1 -- python $R/bench.py --stage 1 --steps 8 --warmup 2 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_s1.log 2>&1
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_c5.log 2>&1
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s3 -- python $R/bench.py --stage 3 --steps 20 --warmup 5 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_s3.log 2>&1
cd $R
for t in s1 c5 s3; do find gpurun_out/prof_$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_${t}_kernel_stats.csv; rm -rf gpurun_out/prof_$t; done
ls -la gpurun_out/r3_*kernel_stats.csv
