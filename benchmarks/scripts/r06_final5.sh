#!/bin/bash
# round-6, after the last test additions: full GPU suite + smoke + the driver command once more (another box)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_final5; mkdir -p $O
ACT_GEMM_TUNE_SAVE=$O/tuned_%p.json python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; grep -v "Warning\|warnings.warn\|pin_memory\|^$" $O/pytest_full.log | tail -4 > $O/pytest.log; cat $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
python bench.py > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.json
