#!/usr/bin/env python3
"""Per-parameter gradient differences HIP vs oracle for one g16 configuration (dev tool): python variant_grad_diff.py noprompt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N, PROMPT_VARIANTS
from act_amd.models import build_model_from_cfg
from act_amd.utils.config import EasyDict
from act_amd.utils.draws import Draws
from oracle import models as OM, layers as OL

tag = sys.argv[1] if len(sys.argv) > 1 else "noprompt"


def noise(shape):
    torch.manual_seed(777)
    return -torch.empty(shape).exponential_().log()


dev = torch.device("cuda:0")
cfg = dict(TINY_STAGE2["dvae_config"]); cfg.update(PROMPT_VARIANTS[tag])
vae = fill_module(build_model_from_cfg(EasyDict(dict(cfg, NAME="ACTPromptedDiscreteVAEwithVIT"))), f"g16.{tag}.").to(dev).train()
if hasattr(vae, "prompt_dropout"):
    vae.prompt_dropout.p = 0.0
pts = torch.from_numpy(clouds(16, TINY_B, TINY_N)).to(dev)
ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": noise((TINY_B, 16, 64))}, device=dev))
lr, lk = vae.get_loss(ret, pts); (lr + 0.1 * lk).backward()
torch.set_num_threads(1)
ora = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg)), f"g16.{tag}.").train(); ora.prompt_p = 0.0
ro = ora(pts.cpu(), OL.Draws({"gumbel": noise((TINY_B, 16, 64))}), temperature=0.7, hard=False)
lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
od = dict(ora.named_parameters())
print(f"loss hip {lr.item():.7f} {lk.item():.7f} oracle {lo[0].item():.7f} {lo[1].item():.7f}")
for i in range(6):
    if torch.is_tensor(ret[i]):
        print("ret", i, (ret[i].detach().cpu() - ro[i].detach()).abs().max().item())
for n, p in vae.named_parameters():
    if p.grad is None or od[n].grad is None:
        continue
    a, r = p.grad.double().cpu(), od[n].grad.double()
    err = (a - r).abs(); scale = max(1.0, r.abs().max().item())
    print(f"{n:40s} max {err.max().item() / scale:.2e} l2 {(err.norm() / r.norm().clamp_min(1e-30)).item():.2e} |ref| {r.norm().item():.3g} {'BAD' if err.max().item() > 1e-4 * scale else ''}")
