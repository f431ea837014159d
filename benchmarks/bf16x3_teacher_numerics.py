#!/usr/bin/env python3
"""Gate 2 of the split-bf16 teacher question (round-4 review item 9), answered on the CPU with the oracle: if the Linear layers of the frozen teacher's ViT
blocks multiplied (hi + lo) bf16 planes of both operands -- three bf16 products hi*hi + hi*lo + lo*hi accumulated in fp32, i.e. 16 significand bits per
operand instead of 24 -- how far would the teacher FEATURES move from the fp32 oracle?  The parity bar is 1e-4 (max |e| / max |ref|, tests/test_gpu_model.py).
Emulation: operands rounded to bf16 planes, each plane product evaluated by an fp32 matmul (products of two 8-bit significands are exact in fp32; the
accumulation is fp32 like the MFMA's).  Also reports bf16x6 (three planes, six products) for reference.  No GPU needed."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.nn.functional as F
from oracle import models as OM, layers as OL
from act_amd.utils.config import cfg_from_yaml_file
from tests.golden.fill import clouds

B = int(os.environ.get("B", "8"))
torch.set_num_threads(min(32, os.cpu_count() or 1))
cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml").model
cfg.dvae_config.ckpt = "none"
torch.manual_seed(2)
model = OM.ACT_PointDistillation(OM.edict(cfg)).train()
tok = model.dvae_tokenizer
pts = torch.from_numpy(clouds(6, B, 1024))
with torch.no_grad():
    nb, center = model.group_divider(pts)
    rec = OL.Draws(record=True)
    ref = tok.forward_tokenizer_features(nb, center, rec)


def planes(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.bfloat16().float(); out.append(p); r = r - p
    return out


def make_linear(n_planes, products):
    def lin(x, w, b=None):
        xs, ws = planes(x, n_planes), planes(w, n_planes)
        y = None
        for i, j in products:                                     # smallest terms first
            t = real_linear(xs[i], ws[j])
            y = t if y is None else y + t
        return y if b is None else y + b
    return lin


real_linear = F.linear
vit_linears = [m for blk in tok.visual_embed[0] for m in blk.modules() if isinstance(m, torch.nn.Linear)]
print(f"B={B}: {len(vit_linears)} Linear layers in the {len(tok.visual_embed[0])} ViT blocks of the teacher; reference features max |x| = {ref.abs().max():.3f}")
for name, n_planes, products in (("bf16x3 (hi+lo, 3 products)", 2, [(1, 0), (0, 1), (0, 0)]),
                                 ("bf16x6 (hi+mid+lo, 6 products)", 3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]),
                                 ("plain bf16 (1 product)", 1, [(0, 0)])):
    lin = make_linear(n_planes, products)
    saved = {}
    for m in vit_linears:
        saved[m] = m.forward
        m.forward = (lambda mm: (lambda x: lin(x, mm.weight, mm.bias)))(m)
    try:
        with torch.no_grad():
            got = tok.forward_tokenizer_features(nb, center, OL.Draws(rec.table))
    finally:
        for m, f in saved.items():
            m.forward = f
    e = (got - ref).abs()
    print(f"  {name:32s} max|e|/max|ref| = {e.max() / ref.abs().max():.3e}   ||e||/||ref|| = {e.norm() / ref.norm():.3e}   (bar: 1e-4)")
