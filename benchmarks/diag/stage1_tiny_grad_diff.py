#!/usr/bin/env python3
"""HIP vs oracle gradients of the tiny Stage-I graph (tests/test_gpu_model.py::test_stage1_tiny_golden), every parameter: max abs difference relative to
max(1, max |ref|), elements beyond 1e-4, and the forward outputs.  Dev diagnostic for kernel changes that move rounding (GELU form, softmax form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N
from act_amd.models import build_model_from_cfg
from act_amd.utils.config import EasyDict
from act_amd.utils.draws import Draws
from oracle import models as OM, layers as OL

def noise(shape, seed=777):
    torch.manual_seed(seed)
    return -torch.empty(shape).exponential_().log()

SEEDS = [int(v) for v in os.environ.get("DIAG_SEEDS", "777").split(",")]
VERBOSE = len(SEEDS) == 1
dev = torch.device("cuda")
for SEED in SEEDS:
  torch.manual_seed(0)
  cfg = EasyDict(TINY_STAGE2["dvae_config"]); cfg.NAME = "ACTPromptedDiscreteVAEwithVIT"
  vae = fill_module(build_model_from_cfg(cfg), "g7.").to(dev).train(); vae.prompt_dropout.p = 0.0
  pts = torch.from_numpy(clouds(4 + (SEED - 777), TINY_B, TINY_N)).to(dev)
  ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": noise((TINY_B, 16, 64), SEED)}, device=dev))
  lr, lk = vae.get_loss(ret, pts); (lr + 0.1 * lk).backward()
  torch.manual_seed(0)
  ora = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg)), "g7.").train(); ora.prompt_p = 0.0
  ro = ora(pts.cpu(), OL.Draws({"gumbel": noise((TINY_B, 16, 64), SEED)}), temperature=0.7, hard=False)
  lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
  print("loss hip", lr.item(), lk.item(), "oracle", lo[0].item(), lo[1].item())
  for i, nm in ((2, "coarse"), (3, "fine"), (5, "logits"), (1, "whole_fine")):
      a, b = ret[i].detach().double().cpu(), ro[i].detach().double()
      print(f"forward {nm:10s} max abs diff {(a - b).abs().max().item():.3e}")
  od = dict(ora.named_parameters()); bad = 0
  for n, p in vae.named_parameters():
      if p.grad is None or od[n].grad is None: continue
      a, b = p.grad.detach().double().cpu(), od[n].grad.double()
      d = (a - b).abs(); rel = (d.max() / max(1.0, b.abs().max())).item(); nb = int((d > 1e-4 * max(1.0, b.abs().max().item())).sum())
      l2 = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
      if rel > 3e-5 and VERBOSE: print(f"{n:50s} rel {rel:.3e}  elements beyond 1e-4: {nb:5d} of {a.numel():7d}   ||e||/||r|| {l2:.3e}")
      bad += rel > 1e-4
  print("seed", SEED, "parameters beyond 1e-4:", int(bad), " worst rel %.3e (%s)" % max(((( p.grad.detach().double().cpu() - od[n].grad.double()).abs().max() / max(1.0, od[n].grad.abs().max().item())).item(), n) for n, p in vae.named_parameters() if p.grad is not None and od[n].grad is not None))
