#!/bin/bash
# round-6: three consecutive runs of the GPU suite at HEAD (flakiness check of the restated Stage-I tiny gradient test and the new GELU tests)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_suite3; mkdir -p $O
for i in 1 2 3; do python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/runs.txt; done
