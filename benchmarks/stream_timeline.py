#!/usr/bin/env python3
"""Per-stream timeline of the production (overlapped, pipelined) Stage-II step from hipEvents -- no profiler attached, so the host is as fast as in
bench.py (rocprofv3's kernel trace makes the ~500 launches per step host-bound and distorts exactly what this measures).

Events on the main stream: step start, after forward (loss enqueued), after backward, after the optimizer; on the teacher stream (auxiliary stream 0):
before / after the grouping + frozen-teacher forward of the next batch.  Prints, per step and averaged, when each phase starts and ends relative to
the step start, i.e. which stream waits for which."""
import os, sys, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_pretrain import freeze_unused_heads, train_transforms, _Single, _Announced
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.utils.logger import get_logger
import act_amd.kernels as K
for n in ("ACT", "Transformer"):
    get_logger(n).setLevel(logging.ERROR)
import bench

cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml"); cfg.model.dvae_config.ckpt = "none"
if os.environ.get("DEPTH"):                                 # what-if: a shallower student (how much of the step do its small kernels really cost?)
    cfg.model.transformer_config.depth = int(os.environ["DEPTH"])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model_from_cfg(cfg.model); freeze_unused_heads(model); model.to(dev).train()
w = _Single(model); opt, _ = builder.build_opti_sche(w, cfg)
pool = [bench.synthetic_clouds(128, 1024, 1 + i, dev) for i in range(4)]
main = torch.cuda.current_stream(dev)
side = K.side_stream(dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
ORDER = os.environ.get("ORDER", "fwd-prefetch-bwd")       # what runner_pretrain.train_step does; "prefetch-first": announce the next batch BEFORE this forward

nxt = None
def step(i, rec):
    global nxt
    cur = nxt if nxt is not None else train_transforms(pool[i % 4].clone())
    nxt = pool[(i + 1) % 4].clone()
    e = {k: ev() for k in ("start", "fwd", "bwd", "opt", "t0", "t1")}
    e["start"].record(main)

    def prefetch():
        global nxt
        nxt = train_transforms(nxt)
        _Announced.mark(model, nxt)
        side.wait_stream(main)
        e["t0"].record(side)
        model.prefetch_teacher(nxt)
        e["t1"].record(side)
    if ORDER == "prefetch-first":
        prefetch()
    loss = w(cur)
    e["fwd"].record(main)
    if ORDER != "prefetch-first":
        prefetch()
    loss.backward()
    e["bwd"].record(main)
    opt.step(); opt.zero_grad(set_to_none=True)
    e["opt"].record(main)
    rec.append(e)

rec = []
for i in range(6):
    step(i, rec)
torch.cuda.synchronize()
rec = []
t_wall0 = None
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 16
for i in range(N):
    step(i, rec)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
base = rec[0]["start"]
T = lambda e: base.elapsed_time(e)
print(f"order = {ORDER}; wall per step {wall:.2f} ms")
print(" step |  start |  fwd end (loss) |  bwd end |  opt end || teacher start | teacher end   (ms relative to this step's start; teacher = NEXT batch)")
acc = [0.0] * 5
for i, e in enumerate(rec):
    s = T(e["start"])
    vals = [T(e["fwd"]) - s, T(e["bwd"]) - s, T(e["opt"]) - s, T(e["t0"]) - s, T(e["t1"]) - s]
    if i >= 2:
        acc = [a + v for a, v in zip(acc, vals)]
    print(f" {i:4d} | {s:7.2f} | {vals[0]:8.2f} | {vals[1]:8.2f} | {vals[2]:8.2f} || {vals[3]:8.2f} | {vals[4]:8.2f}")
n = len(rec) - 2
print(" mean |         | " + " | ".join(f"{a / n:8.2f}" for a in acc[:3]) + " || " + " | ".join(f"{a / n:8.2f}" for a in acc[3:]))
