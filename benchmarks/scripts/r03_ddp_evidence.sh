# round 3, VERDICT item 1: the RCCL path on the one GPU of the box + eight enqueue loops on one host.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q > gpurun_out/r03_ddp_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_ddp_pytest.log
tail -5 gpurun_out/r03_ddp_pytest.log
# (a) the driver's line, then the same through DDP / RCCL at world size 1 (same seeds -> same loss)
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_c2.json 2> gpurun_out/r03_bench_c2.err
ACT_BENCH_FORCE_DDP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_ddp1.json 2> gpurun_out/r03_bench_ddp1.err
# (b) kernel trace of the DDP run: what RCCL launches at world size 1
cd /tmp; export TMPDIR=/tmp
ACT_BENCH_FORCE_DDP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ddp1 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-instrument > $R/gpurun_out/prof_ddp1.log 2>&1
cd $R
find gpurun_out/prof_ddp1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_ddp1_kernel_stats.csv
rm -rf gpurun_out/prof_ddp1
# (c) eight ranks on this host, all on cuda:0 (gloo): per-rank host enqueue cost when 8 Python loops share the cores
ACT_BENCH_SHARE_GPU=1 ACT_BENCH_BACKEND=gloo ACT_GEMM_AUTOTUNE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 8 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-instrument > gpurun_out/r03_bench_8ranks_shared.json 2> gpurun_out/r03_bench_8ranks_shared.err
tail -c 600 gpurun_out/r03_bench_8ranks_shared.json
nproc; python -c "import os; print(os.cpu_count())"
