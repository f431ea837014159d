#!/bin/bash
# three consecutive runs of the GPU suite (product defaults) + smoke
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do python -m pytest tests -q -m gpu 2>&1 | tail -1; done > gpurun_out/r04_gpu_suite_3_runs.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r04_gpu_suite_3_runs.txt
cat gpurun_out/r04_gpu_suite_3_runs.txt
