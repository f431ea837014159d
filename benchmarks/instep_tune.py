#!/usr/bin/env python3
"""IN-STEP choice of the GEMM launch configuration (tile id, split-K) for the biggest products of the production Stage-II step.

The shipped table (act_amd/gemm_tune_gfx950.json) holds the winners of ISOLATED timings; in the overlapped step two or three streams share the chip, and a
tile that wins alone (more LDS, more waves) can lose there (round 4 found that for two shapes by hand).  This script does the comparison where it
counts: coordinate descent over the largest products of the step -- for each shape, every candidate configuration is registered with both host paths
(kernels._GEMM_CACHE and the C-side table of the composites), the overlapped step is timed (hipEvents over windows of steps, interleaved with the
incumbent), and a candidate replaces the incumbent only if it wins by more than the measurement noise, twice.

    python benchmarks/instep_tune.py [--stage 2] [--top 24] [--window 24] > gpurun_out/instep_tune.txt
Prints the accepted changes as JSON (key -> [tile, split]) for benchmarks/merge_tuned.py / a hand edit of the table."""
import os, sys, json, logging, argparse, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import act_amd.kernels as K
import act_amd.composite as CP
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_pretrain import freeze_unused_heads, train_transforms, _Single, _Announced
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.utils.logger import get_logger
for n in ("ACT", "Transformer"):
    get_logger(n).setLevel(logging.ERROR)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=24)
ap.add_argument("--window", type=int, default=24)
ap.add_argument("--margin", type=float, default=0.04, help="ms per step a candidate must win by")
args = ap.parse_args()

seen = {}
_orig = CP._tune_shape


def _spy(ak, bk, M, N, Kd, device):
    seen[(int(ak), int(bk), M, N, Kd)] = seen.get((int(ak), int(bk), M, N, Kd), 0) + 1
    return _orig(ak, bk, M, N, Kd, device)


CP._tune_shape = _spy

cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml"); cfg.model.dvae_config.ckpt = "none"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model_from_cfg(cfg.model); freeze_unused_heads(model); model.to(dev).train()
w = _Single(model); opt, _ = builder.build_opti_sche(w, cfg)
pool = [bench.synthetic_clouds(128, 1024, 1 + i, dev) for i in range(4)]
main = torch.cuda.current_stream(dev)
side = K.side_stream(dev)
state = {"nxt": None, "i": 0}


def step():
    i = state["i"]; state["i"] += 1
    cur = state["nxt"] if state["nxt"] is not None else train_transforms(pool[i % 4].clone())
    nxt = pool[(i + 1) % 4].clone()
    loss = w(cur)
    nxt = train_transforms(nxt)
    _Announced.mark(model, nxt)
    side.wait_stream(main)
    model.prefetch_teacher(nxt)
    state["nxt"] = nxt
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)


def window(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n):
        step()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(8):
    step()
torch.cuda.synchronize()
for key in K._GEMM_CACHE:                                      # products launched through the Python path
    seen.setdefault(tuple(key[:5]), 1)


def current(key):
    c = K._GEMM_CACHE.get(key + (dev.index,)) or K._GEMM_TABLE.get(key)
    return tuple(c) if c else None


def apply(key, cfgv):
    K._GEMM_CACHE[key + (dev.index,)] = tuple(cfgv)
    CP.lib.act_gemm_tune_set(*key, int(cfgv[0]), int(cfgv[1]))


def candidates(key):
    ak, bk, M, N, Kd = key
    sp_list = [1] + [s for s in (2, 3, 4, 6, 8) if Kd // s >= 256 and -(-M // 64) * -(-N // 64) * s <= 8192 and -(-M // 128) * -(-N // 128) < 512]
    if ak and bk:
        tiles = [t for t, bn in ((30, 128), (31, 64), (32, 64), (10, 128), (11, 64), (12, 64), (20, 128), (21, 64)) if N % bn == 0 and (t < 20 or M % 128 == 0)]
    elif ak and not bk:
        tiles = [t for t, bn in ((33, 128), (34, 128), (36, 64), (35, 64)) if N % bn == 0]
    elif not ak and not bk and M % 128 == 0 and N % 128 == 0:
        tiles = [33, 13]
    else:
        tiles = []
    return [(t, s) for t in tiles for s in sp_list]


ranked = sorted((k for k in seen if k[2] * k[3] * k[4] >= (1 << 28)), key=lambda k: -k[2] * k[3] * k[4])[:args.top]
print(f"{len(seen)} GEMM shapes seen in the step; tuning the {len(ranked)} largest in-step", flush=True)
base = statistics.median(window(args.window) for _ in range(3))
print(f"incumbent table: {base:.3f} ms / step", flush=True)
accepted = {}
for key in ranked:
    inc = current(key)
    if inc is None:
        continue
    ws = K.workspace(dev)
    rows = []
    for c in candidates(key):
        if c == inc or c[1] * key[2] * key[3] * 4 > ws.numel() * 4:
            continue
        apply(key, c)
        try:
            step(); torch.cuda.synchronize()
        except Exception as ex:                                  # a configuration the kernel family rejects for this shape
            apply(key, inc)
            rows.append((c, None)); continue
        t = window(args.window)
        apply(key, inc)
        tb = window(args.window)
        rows.append((c, t - tb))
    rows = [(c, d) for c, d in rows if d is not None]
    if not rows:
        continue
    best, d = min(rows, key=lambda r: r[1])
    line = f"{key} incumbent {inc}: " + "  ".join(f"{c}:{dd:+.3f}" for c, dd in sorted(rows, key=lambda r: r[1])[:5])
    if d < -args.margin:                                         # confirm: two more interleaved pairs
        ds = []
        for _ in range(2):
            apply(key, best); t = window(args.window)
            apply(key, inc); tb = window(args.window)
            ds.append(t - tb)
        line += f"   confirm {best}: {ds[0]:+.3f} {ds[1]:+.3f}"
        if max(ds) < -args.margin / 2:
            apply(key, best)
            accepted[",".join(str(v) for v in key)] = list(best)
            line += "  ACCEPTED"
    print(line, flush=True)
final = statistics.median(window(args.window) for _ in range(3))
print(f"after: {final:.3f} ms / step (before {base:.3f})")
print("ACCEPTED " + json.dumps(accepted))
