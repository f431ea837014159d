# ten consecutive runs of the GPU suite with the product default (first-use autotuner ON, complete shipped table)
cd $GRAFT_REPO_ROOT
: > gpurun_out/r03_suite10.txt
for i in $(seq 1 ${RUNS:-10}); do
  timeout 1500 python -m pytest tests/ -q -m gpu > gpurun_out/r03_suite10_run$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/r03_suite10_run$i.log)" >> gpurun_out/r03_suite10.txt
  grep -E "^FAILED|auto-tuned during" gpurun_out/r03_suite10_run$i.log >> gpurun_out/r03_suite10.txt
  [ $i -gt 1 ] && rm -f gpurun_out/r03_suite10_run$i.log
done
cat gpurun_out/r03_suite10.txt
