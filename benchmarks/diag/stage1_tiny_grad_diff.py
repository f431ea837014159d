#!/usr/bin/env python3
"""Where does the 1.15e-4 gradient-norm difference of the tiny Stage-I golden (g7) come from?  HIP path vs CPU oracle on the same
inputs / draws: per-parameter elementwise error distribution + agreement of the discrete selections (Chamfer arg-min)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N
from oracle import models as OM, layers as OL
from act_amd.models import build_model_from_cfg
from act_amd.utils.config import EasyDict
from act_amd.utils.draws import Draws

dev = torch.device("cuda:0")
cfg = dict(TINY_STAGE2["dvae_config"]); cfg["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
torch.manual_seed(0)
vae = fill_module(build_model_from_cfg(EasyDict(cfg)), "g7.").to(dev).train(); vae.prompt_dropout.p = 0.0
ora = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg)), "g7.").train(); ora.prompt_p = 0.0
torch.manual_seed(777); noise = -torch.empty((TINY_B, 16, 64)).exponential_().log()
pts = torch.from_numpy(clouds(4, TINY_B, TINY_N))
ro = ora(pts, OL.Draws({"gumbel": noise}), temperature=0.7, hard=False); lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
rg = vae(pts.to(dev), temperature=0.7, hard=False, draws=Draws({"gumbel": noise}, device=dev)); lg = vae.get_loss(rg, pts.to(dev)); (lg[0] + 0.1 * lg[1]).backward()
print("loss", [x.item() for x in lo], [x.item() for x in lg])
for i in (2, 3, 5):
    d = (rg[i].detach().cpu() - ro[i].detach()).abs()
    print("ret", i, "max abs diff", d.max().item(), "ref max", ro[i].abs().max().item())
od = dict(ora.named_parameters())
for n, p in vae.named_parameters():
    if p.grad is None or od[n].grad is None: continue
    g, r = p.grad.detach().cpu().double().flatten(), od[n].grad.double().flatten()
    e = (g - r).abs(); sc = max(1.0, r.abs().max().item())
    bad = (e > 1e-4 * sc).sum().item()
    print(f"{n:40s} norm {g.norm().item():.6f} vs {r.norm().item():.6f}  rel-norm-diff {abs(g.norm()-r.norm()).item()/max(1,r.norm().item()):.2e}  "
          f"max-elt-err/scale {e.max().item()/sc:.2e}  elts>1e-4: {bad}/{e.numel()}  ||e||/||r|| {e.norm().item()/r.norm().item():.2e}")
