#!/usr/bin/env python3
"""bench.py -- Stage-II ACT pretraining step throughput on MI355X (BASELINE.json metric, configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = augmentation + forward + backward + AdamW update of ACT_PointDistillation on a synthetic batch of
B=128 clouds x 1024 points per GPU (64 groups x 32 neighbours, 12-layer d=384 student, frozen ViT-B teacher),
fp32, random weights, inputs resident in HBM.  Rank 0 prints ONE JSON line: whole-job clouds/s, plus
  roofline      live hipEvent timing of the dominant HIP kernel (instrumented pass after the timed region)
  cpu_baseline  the CPU oracle (pure-PyTorch restatement of the reference path) timed on the host cores (N=1 only)
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
    os.environ["NCCL_DEBUG"] = "WARN"               # keep RCCL's version banner off stdout (one JSON line contract)
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # ... and whatever RCCL still has to say (e.g. its 'Missing "iommu=pt"' warning at init) goes to stderr

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (only priced against in the opt-in split-bf16 teacher configuration)
PEAK_HBM_GBS = 8000.0               # HBM3E spec (6290 GB/s measured-achievable)


def synthetic_clouds(B, N, seed, device):
    """pc_norm'd gaussian clouds (datasets/ShapeNet55Dataset.py:45-51 semantics), generated on the host once."""
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(B, N, 3, generator=g)
    pts = pts - pts.mean(dim=1, keepdim=True)
    pts = pts / pts.norm(dim=2).max(dim=1)[0].view(B, 1, 1)
    return pts.to(device)


def _time_steps(step, budget_s, max_n, min_n=2):
    step()                                             # warm-up
    t0 = time.time(); n = 0
    while n < min_n or (time.time() - t0 < budget_s and n < max_n):
        step(); n += 1
    return (time.time() - t0) / n, n


def cpu_c1_reference(threads, warm=2, runs=7):
    """BASELINE configs[0] / BASELINE.md section 3 'C1': ONE batch of 4 x 1024 x 3 pc-normalised clouds -> Group (FPS 64, kNN 32) ->
    mini-PointNet Encoder(128) -> 2-layer d=128 Transformer encoder (2 heads x 64), forward, PyTorch CPU; median of 7 after 2 warm-ups."""
    from oracle import models as OM, layers as OL
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    grp, enc = OM.Group(64, 32), OL.Encoder(128).train()
    blocks = OL.TransformerEncoder(128, 2, 2, 0.0).train()
    pos = torch.nn.Sequential(torch.nn.Linear(3, 128), torch.nn.GELU(), torch.nn.Linear(128, 128))
    pts = synthetic_clouds(4, 1024, 5, "cpu")
    t_all, t_grp = [], []
    with torch.no_grad():
        for i in range(warm + runs):
            t0 = time.perf_counter()
            nb, center = grp(pts)
            t1 = time.perf_counter()
            blocks(enc(nb), pos(center), OL.Draws())
            t2 = time.perf_counter()
            if i >= warm:
                t_all.append(t2 - t0); t_grp.append(t1 - t0)
    med = sorted(t_all)[len(t_all) // 2]; medg = sorted(t_grp)[len(t_grp) // 2]
    return {"ms": 1e3 * med, "clouds_per_s": 4 / med, "group_Mpts_per_s": 4 * 1024 / medg / 1e6, "threads": threads, "runs": len(t_all),
            "sample": "configs[0]: 4 x 1024 x 3 clouds -> Group(FPS 64, kNN 32; numpy oracle) -> Encoder(128) -> 2-layer d=128 encoder, forward, "
                      f"median of {runs} after {warm} warm-up(s)"}


def cpu_baseline(cfg_model, stage=2, c5=False, seconds_budget=22.0):
    """One training step (fwd+bwd+AdamW) of the CPU oracle on the host cores, bounded sample: Stage II at B=16, Stage I at B=8
    (BASELINE.md section 3, C3-cpu), the C5 stress geometry at B=2.  Timed with ACT_CPU_BASELINE_THREADS (default 32) threads AND with every
    visible core (os.cpu_count()); ``value`` is the faster of the two, ``cores`` the thread count it used, both timings are listed."""
    from oracle import models as OM, layers as OL
    ncpu = os.cpu_count() or 1
    few = min(ncpu, int(os.environ.get("ACT_CPU_BASELINE_THREADS", "32")))
    torch.manual_seed(0)
    if stage == 1:
        B, N, what = 8, 1024, "Stage-I autoencoder steps (tokenizer + prompted ViT-B + FoldingNet, Chamfer-L1 + KL, fwd+bwd+AdamW)"
        model = OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg_model)).train()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.05)
    else:
        B, N = (2, 8192) if c5 else (16, 1024)
        what = "Stage-II steps (fwd+bwd+AdamW)" + (" at the configs[4] stress geometry" if c5 else "")
        model = OM.ACT_PointDistillation(OM.edict(cfg_model)).train()
        opt = torch.optim.AdamW(OM.param_groups(model, 0.05), lr=1e-3, weight_decay=0.05)
    pts = synthetic_clouds(B, N, 99, "cpu")

    def step():
        if stage == 1:
            ret = model(pts, OL.Draws(), temperature=1.0, hard=False)
            lr_, lk_ = model.get_loss(ret)
            loss = lr_ + 0.1 * lk_
        else:
            loss = model(pts)
        loss.backward()
        opt.step(); opt.zero_grad()
    # thread sweep of the training step: ACT_CPU_BASELINE_THREADS and twice that (one step of this graph on all 256 visible cores of the
    # GPU box takes 73 s -- oversubscribed intra-op pools --, 1.7 s on 32: the all-cores timing is taken on the cheap C1 forward below)
    sweep = {}
    for th, share, cap in ((few, 0.7, 8), (min(ncpu, 2 * few), 0.3, 3)):
        if th in sweep:
            continue
        torch.set_num_threads(th)
        dt, n = _time_steps(step, seconds_budget * share, cap, min_n=2 if th == few else 1)
        sweep[th] = {"clouds_per_s": B / dt, "steps": n, "s_per_step": dt}
    best = max(sweep, key=lambda t: sweep[t]["clouds_per_s"])
    out = {"value": sweep[best]["clouds_per_s"], "unit": "clouds/s", "cores": best, "kind": "port",
           "sample": f"{sweep[best]['steps']} {what} of the pure-PyTorch CPU oracle at B={B}, N={N}, same geometry",
           "host_cores_visible": ncpu, "thread_sweep": {str(k): v for k, v in sweep.items()}}
    torch.set_num_threads(few)
    try:                                               # (before the all-cores run below: 256 spinning intra-op threads would starve OpenMP)
        G_, M_ = (512, 64) if c5 else (64, 32)
        out.update(cpu_group_baseline(B=32 if c5 else 128, N=N, G=G_, M=M_))
    except Exception as e:
        out["group_sample"] = f"failed: {e}"
    if stage == 2 and not c5:
        try:
            out["c1_reference"] = cpu_c1_reference(few)
            if ncpu != few:                            # BASELINE.md section 3 asks for os.cpu_count() threads: reported next to the faster setting
                out["c1_reference_all_cores"] = cpu_c1_reference(ncpu, warm=1, runs=3)      # (6.7 s per pass on 256 oversubscribed threads)
        except Exception as e:
            out["c1_reference"] = {"failed": str(e)}
    torch.set_num_threads(few)
    return out


def cpu_group_baseline(B=128, N=1024, G=64, M=32, reps=3):
    """Group (FPS + kNN + gather + centre subtract) of the plain-C oracle (oracle/point_ops_c.c, OpenMP over clouds) on the host cores."""
    import ctypes
    import subprocess
    import numpy as np
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "liboracle_point_ops.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", d, "-s"])
    lib = ctypes.CDLL(so)
    pts = synthetic_clouds(B, N, 7, "cpu").numpy()
    c = np.empty((B, G, 3), np.float32); nb = np.empty((B, G, M, 3), np.float32)
    f = np.empty((B, G), np.int32); k = np.empty((B, G, M), np.int64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_group_f32(P(pts), B, N, G, M, P(c), P(nb), P(f), P(k))
    t0 = time.time()
    for _ in range(reps):
        lib.oracle_group_f32(P(pts), B, N, G, M, P(c), P(nb), P(f), P(k))
    dt = (time.time() - t0) / reps
    return {"group_Mpts_per_s": B * N / dt / 1e6, "group_ms": 1e3 * dt,
            "group_sample": f"Group(FPS {G} + kNN {M}) on {B} x {N} clouds, plain-C oracle, OpenMP over clouds ({os.cpu_count()} host cores visible)"}


def other_workloads(timeout_s=100):
    """python bench.py --stage 1 / --config c5 (10 / 6 timed steps, no instrumented pass, no CPU baseline) -> {name: {clouds_per_s, ms_per_step,
    final_loss, steps, workload}}; a failure is reported in place of the numbers, never raised."""
    import subprocess
    res = {}
    for name, extra, xenv in (("stage1", ["--stage", "1", "--steps", "10", "--warmup", "3"], {"ACT_TEACHER_BF16X3": "0"}),
                              ("c5", ["--config", "c5", "--steps", "6", "--warmup", "2"], {"ACT_TEACHER_BF16X3": "0"}),
                              # NOT the headline: the same configs[1] step with the frozen teacher's ViT products on the split-bf16 kernel (opt-in)
                              ("stage2_teacher_split_bf16_OPT_IN", ["--steps", "20", "--warmup", "5"], {"ACT_TEACHER_BF16X3": "1"}),
                              ("stage1_vit_split_bf16_OPT_IN", ["--stage", "1", "--steps", "10", "--warmup", "3"], {"ACT_TEACHER_BF16X3": "1"})):
        if name.startswith("stage2_") and name.endswith("OPT_IN") and os.environ.get("ACT_TEACHER_BF16X3") == "1":
            continue                                     # the parent already runs that configuration (and says so in its metric and dtype)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--no-instrument", "--no-other-workloads"] + extra
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(xenv)
        t0 = time.time()
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                res[name] = {"failed": (r.stderr or r.stdout)[-300:], "wall_s": time.time() - t0}
                continue
            d = json.loads(line[-1])
            res[name] = {"clouds_per_s": d["value"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "final_loss": d["config"]["final_loss"], "steps": d["steps"],
                         "warmup": d["warmup"], "clouds_per_gpu": d["config"]["clouds_per_gpu"], "metric": d["metric"],
                         "workload": d["config"]["workload"], "wall_s": time.time() - t0}
        except Exception as e:
            res[name] = {"failed": str(e)[-300:], "wall_s": time.time() - t0}
    # the children inherit every ACT_* switch of this process: say which were set, so an A/B run of the parent cannot silently relabel these numbers
    res["act_env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("ACT_")}
    return res


def point_op_latency_models(pts, G, M, C, reps=20, clock_ghz=2.4):
    """What bounds FPS and kNN-group (their HBM fraction is small by construction and must be read next to a latency model).
    fps_chain: FPS is G-1 dependent arg-max steps; per-step latency as launched (hipEvents over the kernel / (G-1)) against the floor = sum of the four
    dependent phases of one step, each measured on its own by the s_memrealtime-stamped micro act_fps_chain_probe (one workgroup of the same launch
    configuration, chip otherwise idle): distance evaluations, wave arg-max, cross-wave arg-max (LDS slot + barrier), winner's coordinates.
    knn_valu: kNN-group is VALU-issue bound: G*N exact-rounded distance evaluations (3 sub + 3 mul + 2 add, no FMA: 8 lane-ops) + G*K extract-min rounds
    (6-step DPP min + ballot + readlane + tournament-tree refresh: ~40 wave instructions) against 4 issue cycles per wave instruction on 1,024 SIMDs."""
    import ctypes
    from act_amd.pointnet2_ops import pointnet2_utils as pu
    from act_amd.knn_cuda import knn_group
    B, N, _ = pts.shape

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3                # us
    _, center = pu.furthest_point_sample_with_centers(pts, G)
    fps_us = timed(lambda: pu.furthest_point_sample_with_centers(pts, G))
    knn_us = timed(lambda: knn_group(pts, center, M, want_nbr=True))
    res = {}
    us = (ctypes.c_double * 5)(); w, p = ctypes.c_int(0), ctypes.c_int(0)
    if N <= 8192:
        rc = C.lib.act_fps_chain_probe(N, 4000, us, ctypes.byref(w), ctypes.byref(p), C.stream())
        if rc != 0:
            raise RuntimeError(f"act_fps_chain_probe -> {rc}")
        floor = sum(us[0:4])
        per_it = fps_us / max(G - 1, 1)
        res["fps_chain"] = {"iterations": G - 1, "kernel_us": fps_us, "us_per_iteration": per_it, "floor_us_per_iteration": floor,
                            "frac_of_floor": floor / per_it,
                            "phases_us": {"distance_evals_%d_per_lane" % p.value: us[0], "wave_argmax_dpp": us[1], "cross_wave_lds_barrier_%d_waves" % w.value: us[2],
                                          "centre_reload_lds": us[3]},
                            "whole_iteration_one_workgroup_alone_us": us[4], "waves": w.value, "points_per_lane": p.value,
                            "note": "floor = sum of the four dependent phases, each timed as 4000 dependent repetitions in one workgroup on an idle chip "
                                    "(s_memrealtime); us_per_iteration = hipEvent time of the launch at B=%d / (G-1), which also carries launch, cloud load and "
                                    "index / centre write-back" % B}
    evals, rounds = float(B) * G * N, float(B) * G * M
    wave_instr = evals / 64.0 * 8.0 + rounds * 40.0
    floor_knn = wave_instr * 4.0 / (1024.0 * clock_ghz * 1e3)   # us: 4 issue cycles per wave instruction, 1,024 SIMDs
    res["knn_valu"] = {"distance_evals": evals, "extract_min_rounds": rounds, "wave_instructions_model": wave_instr, "floor_us": floor_knn, "kernel_us": knn_us,
                       "frac_of_floor": floor_knn / knn_us, "clock_ghz_assumed": clock_ghz,
                       "note": "VALU-issue model: 8 lane-ops per exact-rounded squared distance (packed fp32 halves the mul / add part: the floor is an upper "
                               "bound on the work), ~40 wave instructions per extract-min round; HBM side of this launch: %.1f us" %
                               ((12.0 * N + 12.0 * G + 20.0 * G * M) * B / PEAK_HBM_GBS / 1e3)}
    return res


XGMI_LINK_GBS = 153.0               # per xGMI link and direction; 7 links per GPU (point-to-point, fully connected 8-GPU node)


def self_launch(n, argv):
    """``python bench.py --gpus N`` (N > 1) WITHOUT a launcher environment: spawn the N ranks here -- re-run this same file under
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`` (the driver's own launch form;
    reference main.py:21-28,44-58 reads the ranks from the launcher env the same way) -- and hand its stdout (rank 0's ONE JSON line) and exit code
    through.  One-node only, as the reference (utils/dist_utils.py:9-24: local_rank == rank)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ACT_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: no launcher environment (WORLD_SIZE unset) -- spawning {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, cwd=ROOT, env=env)


def allreduce_probe(device, numel, world, backend, reps=10, warm=3):
    """standalone timing of ONE all-reduce (sum) of the Stage-II gradient payload (``numel`` fp32 = every trainable parameter,
    tools/runner_pretrain.py:89 DDP payload): hipEvents on the current stream, which torch's process group makes wait for the collective's own
    stream.  bus GB/s = 2 (N-1)/N x bytes / t (ring convention); xGMI peak for a ring is ONE link per direction per neighbour (153 GB/s), for a
    fully-connected direct all-reduce 7 links."""
    buf = torch.ones(numel, device=device, dtype=torch.float32)
    for _ in range(warm):
        dist.all_reduce(buf)
        buf.fill_(1.0)
    torch.cuda.synchronize(); dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); dist.all_reduce(buf); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]
    t = torch.tensor([ms], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = t.item()
    nbytes = 4.0 * numel
    alg = nbytes / (ms * 1e-3) / 1e9
    bus = alg * 2.0 * (world - 1) / world
    return {"backend": backend, "world_size_seen_by_rccl": dist.get_world_size(), "allreduce_payload_MB": nbytes / 1e6, "allreduce_ms": ms,
            "allreduce_alg_GBps": alg, "allreduce_bus_GBps": bus,
            "allreduce_GBps_vs_xgmi": {"bus_over_one_link": bus / XGMI_LINK_GBS, "bus_over_seven_links": bus / (7 * XGMI_LINK_GBS),
                                       "link_GBps": XGMI_LINK_GBS},
            "allreduce_note": f"median of {reps} back-to-back all-reduces of the whole gradient payload, max over ranks, nothing else on the chip "
                              "(in the training step the 25 MB DDP buckets overlap the backward); world 1 moves no data"}


_NEXT = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="clouds per GPU (default: 128 for stages 1/2/4, 32 for stage 3)")
    ap.add_argument("--stage", type=int, default=2, choices=(1, 2, 3, 4),
                    help="2: Stage-II distillation step (BASELINE metric, default); 1: Stage-I autoencoder step (configs[2]); "
                         "3: PointTransformer finetune step (finetune_modelnet.yaml: 8192 raw pts -> FPS 1200 -> 1024, fwd+bwd+AdamW); "
                         "4: PointTransformer inference (eval, FPS 8192 -> 1024 + forward)")
    ap.add_argument("--config", default="c2", choices=("c2", "c5"),
                    help="c2: BASELINE configs[1] geometry (default, the driver's line); c5: BASELINE configs[4] stress geometry "
                         "(N=8192 pts, 512 groups x 64 nbrs, 24-layer d=768 student, B=32/GPU) -- stage 2 only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-instrument", action="store_true")
    ap.add_argument("--preflight", action="store_true",
                    help="<= 30 s readiness check of the multi-GPU path before a SCALE run: process-group (RCCL) init on every rank, the all-reduce probe of the "
                         "gradient payload, ONE warm-up + TWO timed DDP steps, per-rank core pinning; one JSON line with preflight.world_size_seen_by_rccl. "
                         "At --gpus 1 a world-1 RCCL group is initialised so the same code path runs")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short configs[2] (Stage I) and configs[4] (C5 stress) timings that the default single-GPU run appends as other_workloads")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: launch the ranks ourselves (a SCALE run must not depend on the caller's launcher)
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < args.gpus and os.environ.get("ACT_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible (ACT_BENCH_SHARE_GPU=1 ACT_BENCH_BACKEND=gloo puts every "
                             "rank on the visible devices round-robin: a test mode, not a measurement)")
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    t_proc0 = time.perf_counter()
    if args.preflight:
        args.steps, args.warmup = 2, 1
        args.no_instrument = args.no_cpu_baseline = args.no_other_workloads = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the product path; there is no CPU fallback)")
    if os.environ.get("ACT_BENCH_SHARE_GPU") == "1":                # testing on a one-GPU box: every rank on cuda:0 (use with ACT_BENCH_BACKEND=gloo)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # one enqueue loop per rank on ONE host: give every rank its own cores, on its GPU's NUMA node where sysfs says which (ACT_PIN_CORES=0: leave it to the OS)
    from act_amd.utils.dist_utils import pin_rank
    affinity = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), local_world, device_index=local_rank)
    force_ddp = os.environ.get("ACT_BENCH_FORCE_DDP") == "1" or args.preflight     # exercise the DDP/RCCL path on a single GPU (testing / preflight)
    t_pg0 = time.perf_counter()
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("ACT_BENCH_BACKEND", "nccl")       # nccl == RCCL over xGMI; gloo only for the shared-GPU test mode
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    pg_init_s = time.perf_counter() - t_pg0
    gpus_flag = args.gpus
    if world != args.gpus:                              # the launcher's environment is the truth (main.py:44-58); never die on the flag
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: reporting n_gpus={world}", file=sys.stderr, flush=True)
        args.gpus = world

    import act_amd._C as C
    from act_amd.models import build_model_from_cfg
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import freeze_unused_heads, train_step, wrap_ddp, _Single
    from act_amd.utils.config import cfg_from_yaml_file
    from act_amd.utils.logger import get_logger
    import logging
    for n in ("ACT", "Transformer"):
        get_logger(n).setLevel(logging.ERROR)

    c5 = args.config == "c5"
    if c5 and args.stage != 2:
        raise SystemExit("--config c5 is the Stage-II stress geometry (use --stage 2)")
    if args.batch is None:
        args.batch = 32 if (args.stage == 3 or c5) else 128
    if args.stage == 2:
        config = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml")
        config.model.dvae_config.ckpt = "none"
        if c5:                                          # BASELINE configs[4]: FPS / kNN in the HBM-bound regime
            tc, dc = config.model.transformer_config, config.model.dvae_config
            tc.embed_dim = tc.encoder_dims = 768; tc.depth = 24; tc.num_heads = tc.decoder_num_heads = 12
            dc.encoder_dims = dc.tokens_dims = dc.decoder_dims = 768
            dc.num_group, dc.group_size = 512, 64
    elif args.stage == 1:
        config = cfg_from_yaml_file("cfgs/autoencoder/act_dvae_with_pretrained_transformer.yaml")
    else:
        config = cfg_from_yaml_file("cfgs/finetune_classification/full/finetune_modelnet.yaml")
    torch.manual_seed(0)                                # identical initial weights on every rank
    model = build_model_from_cfg(config.model)
    freeze_unused_heads(model)
    model.to(device).train()
    ns = argparse.Namespace(local_rank=local_rank, use_gpu=True)
    wrapped = wrap_ddp(model, ns) if (world > 1 or force_ddp) else _Single(model)
    optimizer, _ = builder.build_opti_sche(wrapped, config)
    torch.manual_seed(1234 + rank)                      # per-rank draws (main.py:67 seed + local_rank)

    B, N = args.batch, (8192 if c5 else 1024)
    n_raw = 8192 if args.stage >= 3 else N              # the finetune loaders hand over 8192-point clouds (ModelNet40.yaml)
    pool = [synthetic_clouds(B, n_raw, 1234 + rank * 100 + i, device) for i in range(4)]

    if args.stage == 1:
        from act_amd.tools.runner_autoencoder import train_step as train_step_ae
    if args.stage >= 3:
        from act_amd.tools.runner_finetune import train_step as train_step_ft
        from act_amd.utils import misc
        labels = torch.randint(0, config.model.cls_dim, (B,), device=device)
        if args.stage == 4:
            model.eval()

    def step(i):
        global _NEXT
        if args.stage == 3:                                # look-ahead: the next raw batch is prepared (FPS pool etc.) during this backward
            cur = _NEXT if _NEXT is not None else pool[i % len(pool)].clone()
            _NEXT = pool[(i + 1) % len(pool)].clone()
            return train_step_ft(wrapped, optimizer, cur, labels, config, next_points=_NEXT)[0]
        if args.stage == 4:                                # as runner_finetune.validate: next batch's FPS on the auxiliary stream
            from act_amd.tools.runner_finetune import fps_ahead
            with torch.no_grad():
                cur = _NEXT if _NEXT is not None else fps_ahead(pool[i % len(pool)], N)
                _NEXT = fps_ahead(pool[(i + 1) % len(pool)], N)
                main = torch.cuda.current_stream(device)
                main.wait_event(cur[1])
                cur[0].record_stream(main)
                return model(cur[0]).sum()
        if args.stage == 1:
            l1, l2, _ = train_step_ae(wrapped, optimizer, pool[i % len(pool)], config, 20000 + i)
            return l1 + l2
        # software pipelining across steps: the NEXT batch is handed over too, so its (frozen) teacher forward runs on the
        # auxiliary stream during this batch's backward; every step still executes exactly one teacher forward
        cur = _NEXT if _NEXT is not None else pool[i % len(pool)].clone()
        _NEXT = pool[(i + 1) % len(pool)].clone()
        return train_step(wrapped, optimizer, cur, config, next_points=_NEXT)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    main_prio = int(os.environ.get("ACT_MAIN_PRIO", "0"))           # experiment knob: run the student chain on a high-priority hipStream
    if main_prio:
        hp = torch.cuda.Stream(device=device, priority=main_prio)
        hp.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(hp)
    for i in range(args.warmup):
        step(i)
    barrier()
    WIN = 50                                             # sustained-clock evidence: GPU time per window of 50 steps (hipEvents)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // WIN + 1)] if args.steps >= 2 * WIN else []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if marks and i % WIN == 0:
            marks[i // WIN].record()
        loss = step(i)
    if marks and args.steps % WIN == 0:
        marks[args.steps // WIN].record()
    host_issue = time.perf_counter() - t0                # wall time of the enqueue loop (includes back-pressure waits once the GPU queue is full)
    barrier()
    elapsed = time.perf_counter() - t0
    windows = [marks[j].elapsed_time(marks[j + 1]) / WIN for j in range(len(marks) - 1)] if marks and args.steps % WIN == 0 else []
    # host cost of ONE step against an idle GPU: nothing to wait for, so this is the pure Python + launch time
    idle = []
    for i in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(i)
        idle.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_idle_ms = 1e3 * min(idle)
    host_by_rank = None
    if world > 1:
        hb = [torch.zeros(2, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(hb, torch.tensor([host_idle_ms, 1e3 * host_issue / args.steps], device=device, dtype=torch.float64))
        host_by_rank = [[round(v, 3) for v in t.tolist()] for t in hb]
    affinity_by_rank = [affinity]
    if world > 1:
        affinity_by_rank = [None] * world
        dist.all_gather_object(affinity_by_rank, affinity)
    ms_by_rank = None
    if world > 1:
        el = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(el, torch.tensor([elapsed], device=device, dtype=torch.float64))
        ms_by_rank = [round(1e3 * e.item() / args.steps, 4) for e in el]
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    comm = None
    if world > 1 or force_ddp:
        n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
        comm = allreduce_probe(device, n_train, world, dist.get_backend())
        comm["launched_by"] = "bench.py itself (torch.distributed.run child)" if os.environ.get("ACT_BENCH_SELF_LAUNCHED") == "1" else "external launcher"
        if gpus_flag != world:
            comm["gpus_flag"] = gpus_flag
    loss_val = float(loss.item())
    if not math.isfinite(loss_val):                      # a diverged / NaN step would still be timed happily: refuse to report it
        raise RuntimeError(f"bench: non-finite loss {loss_val} after {args.warmup + args.steps + 3} steps")

    import act_amd.composite as _CP
    x3_on = bool(_CP.TEACHER_BF16X3) and args.stage in (1, 2)          # (Stage I: the products of the same frozen ViT inside the prompt-tuning graph)
    out = {
        "metric": {1: "stage1_autoencoder_point_clouds_per_sec" + ("_vit_split_bf16_opt_in" if x3_on else ""),
                   2: "stage2_pretrain_point_clouds_per_sec" + ("_teacher_split_bf16_opt_in" if x3_on else ""),
                   3: "finetune_cls_point_clouds_per_sec", 4: "inference_cls_point_clouds_per_sec"}[args.stage], "value": B * world * args.steps / elapsed, "unit": "clouds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not x3_on else "f32 + split-bf16 frozen-ViT products (OPT-IN, not the f32 headline)",
        "data": "synthetic",
        "config": {"workload": ("configs[4] stress: ACT Stage-II pretrain step at N=8192 pts, 512 groups x 64 nbrs, 24L d=768 student + 2L decoder, "
                                "frozen 12L ViT-B teacher on 64 prompts + 512 tokens (random init), B=%d clouds/GPU, aug+fwd+bwd+AdamW" % B) if c5 else
                               ("configs[1]: ACT Stage-II pretrain step (pretrain_act_distill.yaml geometry): "
                                "B=128 clouds/GPU x 1024 pts, 64 groups x 32 nbrs, 12L d=384 student + 2L decoder, "
                                "frozen 12L ViT-B teacher (random init), aug+fwd+bwd+AdamW") if args.stage == 2 else
                               ("PointTransformer finetune step (finetune_modelnet.yaml): B=%d clouds/GPU x 8192 raw pts -> FPS 1200 -> "
                                "random 1024, rotate, 64 groups x 32 nbrs, 12L d=384 + mlp-3 head, CE loss, fwd+bwd+clip+AdamW" % B)
                               if args.stage == 3 else
                               ("PointTransformer inference (eval mode): B=%d clouds/GPU x 8192 raw pts -> FPS 1024 -> forward" % B)
                               if args.stage == 4 else
                               ("configs[2]: ACT Stage-I autoencoder step (act_dvae_with_pretrained_transformer.yaml geometry): "
                                "B=%d clouds/GPU x 1024 pts, tokenizer + prompt-tuned frozen ViT-B + FoldingNet, "
                                "Chamfer-L1 + KL losses, fwd+bwd+AdamW" % B),
                   "clouds_per_gpu": B, "points_per_cloud": N, "parallelism": f"dp{world}", "final_loss": loss_val,
                   "host_enqueue_ms_per_step": host_idle_ms,
                   "host_enqueue_note": "wall time to enqueue one whole step against an idle GPU (min of 3): pure host cost; "
                                        "host_loop_ms_per_step is the enqueue loop of the timed region, which includes waiting on a full GPU queue",
                   "host_loop_ms_per_step": 1e3 * host_issue / args.steps,
                   **({"host_ms_per_step_by_rank": host_by_rank,
                       "host_by_rank_note": "[idle-GPU enqueue ms, timed-loop enqueue ms] of every rank (all ranks share this host's cores)"}
                      if host_by_rank else {}),
                   "schedule": ("every timed step = student fwd+bwd+AdamW of batch i on the main stream + grouping and frozen-teacher forward "
                                "of batch i+1 on an auxiliary HIP stream (bit-identical to the sequential schedule; DESIGN section 4)")
                               if args.stage == 2 else "sequential; next batch's FPS prepared on an auxiliary stream (stages 3, 4)",
                   **({"exact_restructurings": ("no term of the reference's arithmetic is dropped: the student's patch embedding runs its last conv + max-pool on the "
                                                "visible 13 of 64 patches (the only tokens MaskTransformer reads) and skips the masked patches' exactly-zero gradient rows -- "
                                                "both bit-identical to computing them (tests/test_gpu_composite.py, test_gpu_model.py); the two products that consume the "
                                                "max-pool gradient walk its one non-zero per (group, channel) instead of a 1/32-dense operand (same terms, fp32 summation "
                                                "order differs from the MFMA order: 1e-6); every layer that feeds a BatchNorm statistic, the whole teacher and the whole "
                                                "student Transformer run in full "
                                                "(DESIGN section 4; ACT_ENCODER_VISIBLE_ONLY=0 ACT_POOL_BWD_LIVE=0 ACT_PN_POOL_BWD_SPARSE=0 restore the dense form)")}
                      if args.stage == 2 else {})},
    }

    if x3_on:
        out["config"]["teacher_products"] = ("OPT-IN ACT_TEACHER_BF16X3=1: the five Linear products of every ViT layer of the FROZEN teacher as hi + lo bf16 planes of both "
                                             "operands, three bf16 MFMA products, fp32 accumulation (csrc/gemm_bf16x3.hip): teacher features move by ~7e-6 of their range "
                                             "(parity bar 1e-4, tests/test_gpu_bf16x3.py); student, losses, gradients and every other teacher kernel are f32-input MFMA")
        if args.stage == 1:
            out["config"]["teacher_products"] = ("OPT-IN ACT_TEACHER_BF16X3=1: the five forward Linear products of every layer of the FROZEN prompt-tuned ViT%s as hi + lo "
                                                 "bf16 planes of both operands, three bf16 MFMA products, fp32 accumulation (csrc/gemm_bf16x3.hip); its LayerNorm / attention "
                                                 "kernels, the dVAE, DGCNNs, FoldingNet and the losses are f32 (block output / gradients move by 3e-6 / <= 1.4e-5: "
                                                 "tests/test_gpu_bf16x3.py)" % (" and the five input-gradient products of its backward" if _CP.TEACHER_BF16X3_BWD else
                                                                                " (backward products f32: ACT_TEACHER_BF16X3_BWD=0)"))
    out["config"]["parity_bar"] = ("loss and features within 1e-4 of the fp32 CPU oracle at this geometry and batch; FPS / kNN indices bit-exact; gradients "
                                   "flip-tolerant: <= 1e-3 of a gradient's elements may exceed 1e-4 (the reference's own fp32 summation-order noise through the "
                                   "hard arg-max / max-pool choices), L2 error <= 5e-3 (tests/test_gpu_model.py)")
    out["config"]["host_affinity_by_rank"] = affinity_by_rank
    if args.preflight:
        out["preflight"] = {"ok": True, "world_size_seen_by_rccl": comm["world_size_seen_by_rccl"], "backend": comm["backend"],
                            "process_group_init_s": round(pg_init_s, 3), "allreduce_ms": comm["allreduce_ms"], "allreduce_payload_MB": comm["allreduce_payload_MB"],
                            "ddp_steps_run": args.warmup + args.steps + 3, "final_loss": loss_val,
                            "stack_chunk": next((m.stack_chunk for m in model.modules() if getattr(m, "stack_chunk", None)), 0),
                            "ranks_pinned": sum(1 for a in affinity_by_rank if a and a.get("pinned")), "wall_s": round(time.perf_counter() - t_proc0, 2),
                            "note": "readiness check only: 2 timed steps are not a measurement (value / ms_per_step on this line are NOT a benchmark result)"}
    if ms_by_rank:
        out["ms_per_step_by_rank"] = ms_by_rank
    if comm:
        out["comm"] = comm
    if windows:
        out["sustained"] = {"window_steps": WIN, "ms_per_step_by_window": windows, "first": windows[0], "last": windows[-1],
                            "note": "hipEvent time of consecutive 50-step windows of the timed region (clock / power drift shows as a slope)"}

    nprof = 3
    if not args.no_instrument:
        # ---- instrumented pass: hipEvents around every launch of libact_hip.so on the launch stream --------------
        # Per-kernel durations are only meaningful when kernels do not share the chip: the auxiliary stream (frozen teacher /
        # weight gradients, which the timed region above runs concurrently with the main chain) is serialised for this pass.
        # EVERY rank runs these steps (each one contains the gradient all-reduce: a rank that skipped them would leave rank 0
        # waiting in a collective); only rank 0 records.
        import act_amd.kernels as KK
        import act_amd.models.act as AM
        saved = (AM._OVERLAP_TEACHER, KK.OVERLAP_DW)
        AM._OVERLAP_TEACHER, KK.OVERLAP_DW = False, False
        step(0); torch.cuda.synchronize()
        if rank == 0:
            C.prof_reset(); C.prof_enable(True)
        for i in range(nprof):
            step(i)
        torch.cuda.synchronize()
        if rank == 0:
            C.prof_enable(False)
        AM._OVERLAP_TEACHER, KK.OVERLAP_DW = saved
    if rank == 0 and not args.no_instrument:
        table = C.prof_table()
        tot_ms = sum(v["ms"] for v in table.values())
        kernels = {}
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            avg_ms = v["ms"] / v["launches"]
            e = {"ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] / nprof, "avg_us": 1e3 * avg_ms}
            if v["flops"] > 0:
                e["tflops"] = v["flops"] / (v["ms"] * 1e-3) / 1e12
            if v["bytes"] > 0:
                e["alg_GBs"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            kernels[k] = e
        dom = next(iter(kernels))
        dv = table[dom]
        if dv["flops"] > 0:
            ach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
            # (opt-in runs only: the split-bf16 kernel executes three bf16 products per algorithmic product -> its ceiling is a third of the dense bf16 peak)
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if dom == "sgemm_nt_bf16x3" else PEAK_F32_MFMA_TFLOPS
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": None,
                               "avg_launch_us": 1e3 * dv["ms"] / dv["launches"], "launches_per_step": dv["launches"] / nprof,
                               "note": "hipEvents per launch, auxiliary stream serialised (ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0); "
                                       "the timed region runs the two streams concurrently"}
        else:
            ach = dv["bytes"] / (dv["ms"] * 1e-3) / 1e9
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": ach / PEAK_HBM_GBS, "traffic": None,
                               "avg_launch_us": 1e3 * dv["ms"] / dv["launches"], "launches_per_step": dv["launches"] / nprof}
        # HBM traffic of the dominant kernel: NOT measured by this run -- PMC counters need rocprofv3 around the process -- but read from the
        # committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command and workload (profiles/rNN_pmc_traffic_<workload>.json,
        # written by benchmarks/pmc_traffic.py); null when no profile of THIS workload is committed.
        import glob
        tag = "c5" if c5 else {1: "s1", 2: "c2", 3: "s3", 4: "s4"}[args.stage]
        out["roofline"]["algorithmic_bytes_per_launch"] = dv["bytes"] / dv["launches"]
        out["roofline"]["traffic_source"] = None
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % tag)), reverse=True):
            try:
                pm = json.load(open(pf))["kernels"].get(dom)
            except Exception:
                pm = None
            if pm:
                out["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = ("committed profile figure, not measured in this run: bytes/launch from separate rocprofv3 "
                                                     "--pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this workload (calibrated), profiles/" + os.path.basename(pf))
                break
        out["kernels"] = kernels
        out["hip_kernel_ms_per_step"] = tot_ms / nprof
        # ---- whole-step roofline: FLOPs the HIP kernels really executed this step (sum of 2mnk of every launch, counted at launch) against the
        # timed region's step time; next to it the reference formulation's count (38.2 GFLOP per cloud for Stage II at the configs[1]
        # geometry) -- clouds/s x 38.2 GFLOP exceeds the fp32 MFMA peak because exact restructurings removed work (config.exact_restructurings).
        ex_tflop = sum(v["flops"] for v in table.values()) / nprof / 1e12
        step_s = elapsed / args.steps
        ref_tflop = {("c2", 2): 38.2e-3 * B, ("c5", 2): 483e-3 * B, ("c2", 1): 70e-3 * B}.get(("c5" if c5 else "c2", args.stage))
        out["roofline"]["step"] = {"executed_tflop": ex_tflop, "reference_formulation_tflop": ref_tflop,
                                   "achieved_tflops": ex_tflop / step_s, "frac": ex_tflop / step_s / PEAK_F32_MFMA_TFLOPS,
                                   "reference_formulation_tflops_equiv": (ref_tflop / step_s) if ref_tflop else None,
                                   "note": "executed = sum over every GEMM / attention launch of one step of 2mnk as launched (per GPU); "
                                           "reference_formulation = SURVEY 8(d) FLOPs of the reference's own graph for this batch"}
        if x3_on:
            out["roofline"]["step"]["note"] += ("; OPT-IN run: the teacher's ViT products (%.2f TFLOP of the step) ran on the bf16 matrix cores as three products each, "
                                                 "so `frac` against the f32-input peak is NOT a utilisation figure here" % (table.get("sgemm_nt_bf16x3", {"flops": 0})["flops"] / nprof / 1e12))
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma_util_%s.json" % tag)), reverse=True):
            try:
                pm = json.load(open(pf))
                mu = {k: round(v["mfma_util_percent_time_weighted"], 1) for k, v in pm["kernels"].items()}
            except Exception:
                continue
            out["roofline"]["mfma_util"] = {"percent_time_weighted": mu, "dominant": mu.get(dom),
                                            "source": "committed profile figure, not measured in this run: rocprofv3 --pmc MfmaUtil pass of this workload, "
                                                      "profiles/" + os.path.basename(pf),
                                            "attention_bound_note": "on gfx950 no VALU instruction of any wave issues while a v_mfma_f32_32x32x2_f32 of the same SIMD executes "
                                                                    "(benchmarks/micro/mfma_valu_overlap.hip, mfma_valu_kinds.hip -> profiles/r06_mfma_valu_*.txt), so an f32 attention "
                                                                    "kernel is bounded by MFMA time PLUS VALU time: forward 64 MFMAs (4,096 cycles) + 104 VALU instructions (~510 cycles) "
                                                                    "per 32-key chunk and wave = 0.89 at best; at the teacher shape (64 q x 128 k) the launch, the un-overlapped HBM fill "
                                                                    "(Q + first chunk of all 768 workgroups) and the drain (O) add ~15 us to 20.5 us of MFMA time "
                                                                    "(profiles/r06_attn_fwd_ablation.txt)"}
            break
        # ---- FPS + kNN "Group" throughput (BASELINE metric part 2), hipEvents on the current stream ------------------
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        grp = model.group_divider if hasattr(model, "group_divider") else None
        gpts = pool[0][:, :N].contiguous()
        for _ in range(3):
            grp(gpts)
        reps = 20
        ev0.record()
        for _ in range(reps):
            grp(gpts)
        ev1.record(); torch.cuda.synchronize()
        gms = ev0.elapsed_time(ev1) / reps
        G_, M_ = (512, 64) if c5 else (64, 32)
        fps_b = (12.0 * N + 16.0 * G_) * B                          # algorithmic bytes / cloud: 13,312 (C2) / 106,496 (C5)
        knn_b = (12.0 * N + 12.0 * G_ + 20.0 * G_ * M_) * B         # 54,016 (C2) / 759,808 (C5)
        out["group_fps_knn"] = {"Mpts_per_s": B * N / (gms * 1e-3) / 1e6, "ms": gms,
                                "alg_GBs": (fps_b + knn_b) / (gms * 1e-3) / 1e9, "frac_hbm": (fps_b + knn_b) / (gms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "frac_hbm_note": "small by construction (BASELINE.md section 2): the %.1f MB of a batch would take %.1f us at the HBM rate; "
                                                 "what bounds these kernels is the serial chain / VALU issue models below" %
                                                 ((fps_b + knn_b) / 1e6, (fps_b + knn_b) / PEAK_HBM_GBS / 1e3)}
        try:
            out["group_fps_knn"].update(point_op_latency_models(gpts, G_, M_, C))
        except Exception as e:                               # a report, never a reason to lose the bench line
            out["group_fps_knn"]["fps_chain"] = {"failed": str(e)[-200:]}

    # ---- BASELINE configs[2] (Stage I, B=128) and configs[4] (C5 stress, B=32): short timings of the same bench in child processes, after the
    # headline's timed region (this process is idle meanwhile) -- so the driver's own line carries them, not only profiles/.  ~20 s each.
    if rank == 0 and world == 1 and args.stage == 2 and not c5 and args.batch == 128 and not args.no_other_workloads:
        out["other_workloads"] = other_workloads()

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.stage in (1, 2):
        try:
            out["cpu_baseline"] = cpu_baseline(config.model, stage=args.stage, c5=c5)
        except Exception as e:                           # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = {"value": None, "unit": "clouds/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if world > 1 or force_ddp:
        dist.barrier()
    if rank == 0:
        try:                                             # RCCL prints its version banner through C stdio: flush it first
            import ctypes                                # so that the JSON line is the last thing on stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
