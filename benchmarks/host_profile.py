#!/usr/bin/env python3
"""cProfile of the host side of the Stage-II step (dev tool): where do the ~26 ms of Python/launch time per step go?"""
import cProfile, pstats, os, sys, argparse, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_pretrain import freeze_unused_heads, train_step, _Single
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.utils.logger import get_logger
for n in ("ACT", "Transformer"):
    get_logger(n).setLevel(logging.ERROR)
import bench
cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml"); cfg.model.dvae_config.ckpt = "none"
dev = torch.device("cuda:0")
model = build_model_from_cfg(cfg.model); freeze_unused_heads(model); model.to(dev).train()
w = _Single(model); opt, _ = builder.build_opti_sche(w, cfg)
pool = [bench.synthetic_clouds(128, 1024, 1 + i, dev) for i in range(4)]
nxt = [None]
def step(i):
    cur = nxt[0] if nxt[0] is not None else pool[i % 4].clone()
    nxt[0] = pool[(i + 1) % 4].clone()
    return train_step(w, opt, cur, cfg, next_points=nxt[0])
for i in range(5): step(i)
torch.cuda.synchronize()
IDLE = "--idle" in sys.argv          # every profiled step starts against an idle GPU: no back-pressure waits inside the numbers
pr = cProfile.Profile()
import time
wall = 0.0
for i in range(10):
    if IDLE:
        torch.cuda.synchronize()
    t0 = time.perf_counter(); pr.enable()
    step(i)
    pr.disable(); wall += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host wall per step ({'idle GPU' if IDLE else 'queued'}): {wall / 10 * 1e3:.2f} ms (with cProfile overhead)")
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumtime").print_stats(45)
