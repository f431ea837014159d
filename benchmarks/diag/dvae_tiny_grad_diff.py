#!/usr/bin/env python3
"""Per-parameter gradient differences HIP vs oracle on the g15 DiscreteVAE tiny geometry (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden.fill import fill_module, clouds, TINY_DVAE, TINY_B, TINY_N
from act_amd.models import build_model_from_cfg
from act_amd.utils.config import EasyDict
from act_amd.utils.draws import Draws
from oracle import models as OM, layers as OL


def noise(shape):
    torch.manual_seed(777)
    return -torch.empty(shape).exponential_().log()


dev = torch.device("cuda:0")
vae = fill_module(build_model_from_cfg(EasyDict(dict(TINY_DVAE))), "g15.").to(dev).train()
pts = torch.from_numpy(clouds(15, TINY_B, TINY_N)).to(dev)
ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": noise((TINY_B, 16, 64))}, device=dev))
lr, lk = vae.get_loss(ret, pts); (lr + 0.1 * lk).backward()
for nt in (1, 8):
    torch.set_num_threads(nt)
    ora = fill_module(OM.DiscreteVAE(OM.edict(TINY_DVAE)), "g15.").train()
    ro = ora(pts.cpu(), OL.Draws({"gumbel": noise((TINY_B, 16, 64))}), temperature=0.7, hard=False)
    lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
    od = dict(ora.named_parameters())
    print(f"--- oracle threads {nt}: loss hip {lr.item():.7f} {lk.item():.7f} oracle {lo[0].item():.7f} {lo[1].item():.7f}")
    for n, p in vae.named_parameters():
        if p.grad is None or od[n].grad is None:
            continue
        a, r = p.grad.double().cpu(), od[n].grad.double()
        err = (a - r).abs(); scale = max(1.0, r.abs().max().item())
        if err.max().item() > 1e-4 * scale:
            bad = (err > 1e-4 * scale)
            rows = bad.reshape(bad.shape[0], -1).any(dim=1).nonzero().flatten().tolist() if bad.dim() > 1 else []
            print(f"{n:40s} max {err.max().item() / scale:.2e} l2 {(err.norm() / r.norm()).item():.2e} bad {int(bad.sum())}/{bad.numel()} rows {rows[:8]} refmax {r.abs().max().item():.3g}")
    if nt == 1:
        keep = {n: g.grad.clone() for n, g in od.items() if g.grad is not None}
    else:
        for n in keep:
            e = (keep[n].double() - od[n].grad.double()).abs(); sc = max(1.0, keep[n].abs().max().item())
            if e.max().item() > 1e-4 * sc:
                print(f"   oracle 1 vs 8 threads {n:40s} max {e.max().item() / sc:.2e} bad {int((e > 1e-4 * sc).sum())}")
