#!/usr/bin/env python3
"""cProfile of the host side of the finetune step (dev tool)."""
import cProfile, pstats, os, sys, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_pretrain import _Single
from act_amd.tools.runner_finetune import train_step
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.utils.logger import get_logger
for n in ("ACT", "Transformer"):
    get_logger(n).setLevel(logging.ERROR)
import bench
cfg = cfg_from_yaml_file("cfgs/finetune_classification/full/finetune_modelnet.yaml")
dev = torch.device("cuda:0")
model = build_model_from_cfg(cfg.model); model.to(dev).train()
w = _Single(model); opt, _ = builder.build_opti_sche(w, cfg)
pool = [bench.synthetic_clouds(32, 8192, 1 + i, dev) for i in range(4)]
labels = torch.randint(0, 40, (32,), device=dev)
def step(i):
    return train_step(w, opt, pool[i % 4], labels, cfg)
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(40)
