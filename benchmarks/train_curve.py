#!/usr/bin/env python3
"""Short end-to-end training runs at the full BASELINE geometry (synthetic data) through the runner's own train_step, to show that
the pipelined Stage-II step and the finetune step optimise: writes loss / accuracy curves as JSON (dev tool, results under profiles/)."""
import json, logging, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import bench
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_pretrain import freeze_unused_heads, train_step, _Single
from act_amd.tools import runner_finetune as RF
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.utils.logger import get_logger
for n in ("ACT", "Transformer"):
    get_logger(n).setLevel(logging.ERROR)
dev = torch.device("cuda:0")
out = {}

# ---- Stage II: 300 steps, B=128, 16 distinct synthetic batches cycled
torch.manual_seed(0)
cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml"); cfg.model.dvae_config.ckpt = "none"
model = build_model_from_cfg(cfg.model); freeze_unused_heads(model); model.to(dev).train()
w = _Single(model); opt, _ = builder.build_opti_sche(w, cfg)
for g in opt.param_groups:
    g["lr"] = 5e-4                                        # constant LR (the schedule's warm-up starts at 1e-6)
pool = [bench.synthetic_clouds(128, 1024, 100 + i, dev) for i in range(16)]
losses, nxt = [], None
for i in range(300):
    cur = nxt if nxt is not None else pool[i % 16].clone()
    nxt = pool[(i + 1) % 16].clone()
    losses.append(train_step(w, opt, cur, cfg, next_points=nxt))
l = torch.stack(losses).cpu().tolist()
out["stage2"] = {"steps": 300, "batch": 128, "lr": 5e-4, "loss_first10_mean": sum(l[:10]) / 10, "loss_last10_mean": sum(l[-10:]) / 10,
                 "loss_every_10": [round(v, 5) for v in l[::10]]}
print("stage2", out["stage2"]["loss_first10_mean"], "->", out["stage2"]["loss_last10_mean"], flush=True)
del model, w, opt, pool

# ---- Stage I: 200 steps, B=128 (full-batch soft gumbel-softmax over 8,192 codes: 2^26 in-kernel noise draws per step)
from act_amd.tools.runner_autoencoder import train_step as train_step_ae
torch.manual_seed(0)
cfg1 = cfg_from_yaml_file("cfgs/autoencoder/act_dvae_with_pretrained_transformer.yaml")
ae = build_model_from_cfg(cfg1.model).to(dev).train()
w1 = _Single(ae); opt1, _ = builder.build_opti_sche(w1, cfg1)
for g in opt1.param_groups:
    g["lr"] = 5e-4
pool = [bench.synthetic_clouds(128, 1024, 300 + i, dev) for i in range(16)]
rec = []
for i in range(200):
    l1, l2, _ = train_step_ae(w1, opt1, pool[i % 16], cfg1, 20000 + i)
    rec.append(torch.stack([l1.detach(), l2.detach()]))
r = torch.stack(rec).cpu().tolist()
assert all(x == x and abs(x) < 1e30 for row in r for x in row), "non-finite Stage-I loss"
out["stage1"] = {"steps": 200, "batch": 128, "lr": 5e-4, "recon_first10_mean": sum(x[0] for x in r[:10]) / 10, "recon_last10_mean": sum(x[0] for x in r[-10:]) / 10,
                 "recon_every_10": [round(x[0], 5) for x in r[::10]], "klv_every_10": [round(x[1], 5) for x in r[::10]]}
print("stage1", out["stage1"]["recon_first10_mean"], "->", out["stage1"]["recon_last10_mean"], flush=True)
del ae, w1, opt1, pool

# ---- finetune: synthetic 40-class ModelNet-shaped data, 150 steps at B=32
from act_amd.datasets import build_dataset_from_cfg
from act_amd.utils.config import EasyDict
cfg3 = cfg_from_yaml_file("cfgs/finetune_classification/full/finetune_modelnet.yaml")
ds = build_dataset_from_cfg(EasyDict(NAME="ModelNet", N_POINTS=8192, NUM_CATEGORY=40, SYNTHETIC=True, NUM_SAMPLES=640, DATA_PATH="none"),
                            EasyDict(subset="train"))
items = [ds[i][2] for i in range(640)]
ft = build_model_from_cfg(cfg3.model); ft.apply(ft._init_weights); ft.to(dev).train()
w3 = _Single(ft); opt3, _ = builder.build_opti_sche(w3, cfg3)
for g in opt3.param_groups:
    g["lr"] = 5e-4
batches = [(torch.stack([items[j][0] for j in range(b * 32, b * 32 + 32)]).to(dev),
            torch.tensor([items[j][1] for j in range(b * 32, b * 32 + 32)], device=dev)) for b in range(20)]
hist = []
for i in range(150):
    pts, lab = batches[i % 20]
    hist.append(torch.stack(RF.train_step(w3, opt3, pts, lab, cfg3, next_points=batches[(i + 1) % 20][0])))
h = torch.stack(hist).cpu().tolist()
out["finetune"] = {"steps": 150, "batch": 32, "classes": 40, "loss_first10_mean": sum(x[0] for x in h[:10]) / 10,
                   "loss_last10_mean": sum(x[0] for x in h[-10:]) / 10, "train_acc_first10_mean": sum(x[1] for x in h[:10]) / 10,
                   "train_acc_last10_mean": sum(x[1] for x in h[-10:]) / 10}
print("finetune", out["finetune"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "train_curves.json"), "w"), indent=1)
