#!/usr/bin/env python3
"""one attention shape, a few launches: target of the rocprofv3 --pmc passes in benchmarks/scripts/attn_pmc.sh (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
B, S0, Sq, H, hd = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 64, 64, 12, 64))]
qkv = torch.randn(B * Sq, 3 * H * hd, device="cuda"); do = torch.randn(B * Sq, H * hd, device="cuda")
kv0 = torch.randn(B * max(S0, 1), 2 * H * hd, device="cuda")
for _ in range(5):
    out, lse = K.attention_fwd_prefix(kv0, S0, qkv, Sq, B, H, hd, want_lse=True)
    K.attention_bwd_prefix(kv0, S0, qkv, Sq, out, do, lse, B, H, hd)
torch.cuda.synchronize()
