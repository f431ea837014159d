set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/r3_c2.json 2> gpurun_out/r3_c2.err
python bench.py --stage 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_s1.json 2>/dev/null
python bench.py --config c5 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r3_c5.json 2>/dev/null
python bench.py --stage 3 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r3_s3.json 2>/dev/null
python bench.py --stage 4 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r3_s4.json 2>/dev/null
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-instrument > gpurun_out/r3_sustained.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3 -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_r3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_r3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > $R/gpurun_out/pmc_fetch_r3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_r3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > $R/gpurun_out/pmc_write_r3.log 2>&1
ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $R/gpurun_out/pmc_mfma_r3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > $R/gpurun_out/pmc_mfma_r3.log 2>&1
cd $R
python benchmarks/pmc_traffic.py gpurun_out/pmc_fetch_r3 gpurun_out/pmc_write_r3 > gpurun_out/r3_pmc_traffic.json
python benchmarks/pmc_mfma_util.py gpurun_out/pmc_mfma_r3 > gpurun_out/r3_pmc_mfma_util.json
find gpurun_out/prof_r3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_kernel_stats.csv
# keep the merged-back payload small: drop the raw traces
rm -rf gpurun_out/prof_r3 gpurun_out/pmc_fetch_r3 gpurun_out/pmc_write_r3 gpurun_out/pmc_mfma_r3
ls -la gpurun_out | tail -15
