#!/bin/bash
# round-5 evidence: the driver's command, a sustained run, per-workload profiles (kernel stats, PMC traffic, MfmaUtil), host profile, self-launched 2-rank line
cd "$GRAFT_REPO_ROOT"
export RND=r05
python bench.py > gpurun_out/r05_bench_driver_cmd.json 2> gpurun_out/r05_bench_driver_cmd.err
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-instrument --no-other-workloads > gpurun_out/r05_bench_sustained.json 2>/dev/null
ACT_BENCH_SHARE_GPU=1 ACT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_bench_selflaunch_2ranks_one_gpu_gloo.json 2> gpurun_out/r05_bench_selflaunch.err
ACT_BENCH_FORCE_DDP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r05_bench_ddp1_rccl.json 2>/dev/null
python benchmarks/host_profile.py --idle > gpurun_out/r05_host_profile_idle.txt 2>&1
bash benchmarks/scripts/profiles.sh c2 > gpurun_out/r05_prof_c2.log 2>&1
bash benchmarks/scripts/profiles.sh s1 --stage 1 > gpurun_out/r05_prof_s1.log 2>&1
bash benchmarks/scripts/profiles.sh c5 --config c5 > gpurun_out/r05_prof_c5.log 2>&1
ls gpurun_out | grep r05_ | head -60
