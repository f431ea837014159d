# eight independent training processes (each its own DDP / RCCL world of 1, its own rendezvous port) sharing cuda:0 and this host's cores:
# the per-process host enqueue cost when eight Python loops coexist (what 8 ranks on one 8-GPU node ask of the host)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/h8
for i in 0 1 2 3 4 5 6 7; do
  ACT_BENCH_FORCE_DDP=1 MASTER_PORT=$((29700 + i)) ACT_GEMM_AUTOTUNE=0 timeout 900 python bench.py --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-instrument \
      > gpurun_out/h8/p$i.json 2> gpurun_out/h8/p$i.err &
done
wait
ACT_BENCH_FORCE_DDP=1 ACT_GEMM_AUTOTUNE=0 timeout 600 python bench.py --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-instrument > gpurun_out/h8/alone.json 2> gpurun_out/h8/alone.err
python - <<'PY'
import json, glob, os
def last(f):
    ls = [l for l in open(f) if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None
out = {"what": "8 concurrent single-rank (DDP over RCCL, world 1) Stage-II training processes at B=16 sharing ONE GPU and this host; host_enqueue = wall time to enqueue one step against an idle GPU (min of 3) -- with 8 processes the GPU is never idle, so host_loop (enqueue loop of the timed region incl. back-pressure) is the relevant upper bound",
       "host_cores": os.cpu_count(), "procs": []}
for f in sorted(glob.glob("gpurun_out/h8/p*.json")):
    d = last(f)
    if d: out["procs"].append({"ms_per_step": d["ms_per_step"], "host_enqueue_ms": d["config"]["host_enqueue_ms_per_step"], "host_loop_ms": d["config"]["host_loop_ms_per_step"], "loss": d["config"]["final_loss"]})
d = last("gpurun_out/h8/alone.json")
out["alone"] = {"ms_per_step": d["ms_per_step"], "host_enqueue_ms": d["config"]["host_enqueue_ms_per_step"], "host_loop_ms": d["config"]["host_loop_ms_per_step"]} if d else None
json.dump(out, open("gpurun_out/r03_host_8procs.json", "w"), indent=1)
print(json.dumps(out))
PY
