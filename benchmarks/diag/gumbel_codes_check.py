#!/usr/bin/env python3
"""soft gumbel-softmax x codebook at the Stage-I sizes: which piece produces non-finite values (dev tool)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import act_amd.kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
cb = torch.randn(8192, 384, device=dev)
for B in (8, 32, 128):
    logits = torch.randn(B, 64, 8192, device=dev) * 2
    y = K.gumbel_softmax(logits, 0.8, noise=None, seed=12345)
    s = y.sum(-1)
    print(f"B={B}: y finite {bool(torch.isfinite(y).all())} rowsum [{s.min().item():.6f}, {s.max().item():.6f}] max {y.max().item():.4f}")
    cbt = cb.t().contiguous()
    out = K.linear(y, cbt, None)
    ref = y.reshape(-1, 8192) @ cb
    print(f"      linear finite {bool(torch.isfinite(out).all())} err {(out.reshape(-1, 384) - ref).abs().max().item():.3e}  cfg {K._GEMM_CACHE.get((1, 1, B * 64, 384, 8192, 0))}")
    g = -torch.empty_like(logits).exponential_().log()
    y2 = K.gumbel_softmax(logits, 0.8, noise=g)
    r2 = torch.softmax((logits + g) / 0.8, -1)
    print(f"      with injected noise: finite {bool(torch.isfinite(y2).all())} err {(y2 - r2).abs().max().item():.3e}")
