cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r03_fullsuite.log 2>&1
tail -15 gpurun_out/r03_fullsuite.log
