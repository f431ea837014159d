#!/bin/bash
# round-6 evidence: the driver's command, a sustained run, the preflight lines, DDP world-1 with / without chunked stacks, per-workload profiles (kernel stats, PMC traffic, MfmaUtil)
cd "$GRAFT_REPO_ROOT"
export RND=r06
python bench.py > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-instrument --no-other-workloads > gpurun_out/r06_bench_sustained.json 2>/dev/null
ACT_BENCH_SHARE_GPU=1 ACT_BENCH_BACKEND=gloo python bench.py --gpus 2 --preflight > gpurun_out/r06_preflight_2ranks_one_gpu_gloo.json 2> gpurun_out/r06_preflight_2ranks.err
python bench.py --gpus 1 --preflight > gpurun_out/r06_preflight_1rank_rccl.json 2> gpurun_out/r06_preflight_1rank.err
for c in 0 4 0 4; do ACT_BENCH_FORCE_DDP=1 ACT_BLOCK_STACK_CHUNK=$c python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-other-workloads --no-instrument | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ddp world 1 (RCCL), ACT_BLOCK_STACK_CHUNK=$c:', d['ms_per_step'], 'ms/step', d['value'], 'clouds/s, final_loss', d['config']['final_loss'])"; done > gpurun_out/r06_ddp1_stack_chunk_cost.txt 2>&1
python bench.py --stage 3 --steps 30 --warmup 8 > gpurun_out/r06_bench_s3.json 2>/dev/null
python bench.py --stage 4 --steps 30 --warmup 8 > gpurun_out/r06_bench_s4.json 2>/dev/null
bash benchmarks/scripts/profiles.sh c2 > gpurun_out/r06_prof_c2.log 2>&1
bash benchmarks/scripts/profiles.sh s1 --stage 1 > gpurun_out/r06_prof_s1.log 2>&1
bash benchmarks/scripts/profiles.sh c5 --config c5 > gpurun_out/r06_prof_c5.log 2>&1
ls gpurun_out | grep r06_ | head -60
