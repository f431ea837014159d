#!/bin/bash
# round-6 closing evidence at HEAD: full GPU suite + smoke, the driver command, Stage-II / Stage-I / C5 profiles (kernel stats, PMC traffic, MfmaUtil)
cd "$GRAFT_REPO_ROOT"; export RND=r06; O=gpurun_out/r06_final4; mkdir -p $O
ACT_GEMM_TUNE_SAVE=$O/tuned_%p.json python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; grep -v "Warning\|warnings.warn\|pin_memory\|^$" $O/pytest_full.log | tail -12 > $O/pytest.log; grep "conftest\]" $O/pytest_full.log | cut -c1-400 >> $O/pytest.log; cat $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
python bench.py > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
bash benchmarks/scripts/profiles.sh c2 > gpurun_out/r06_prof_c2.log 2>&1
bash benchmarks/scripts/profiles.sh s1 --stage 1 > gpurun_out/r06_prof_s1.log 2>&1
bash benchmarks/scripts/profiles.sh c5 --config c5 > gpurun_out/r06_prof_c5.log 2>&1
ls gpurun_out | grep r06_ | head -40
