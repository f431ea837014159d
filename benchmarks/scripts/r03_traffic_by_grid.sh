R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pw -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > /dev/null 2>&1
cd $R
head -1 $(find gpurun_out/pf -name "*counter_collection.csv" | head -1)
python benchmarks/pmc_traffic_by_grid.py gpurun_out/pf gpurun_out/pw > gpurun_out/r03_c2_traffic_by_grid.txt
rm -rf gpurun_out/pf gpurun_out/pw
cat gpurun_out/r03_c2_traffic_by_grid.txt
