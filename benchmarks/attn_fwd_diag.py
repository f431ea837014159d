#!/usr/bin/env python3
"""teacher-shape attention forward under the ablation variants of ACT_ATTN_FWD_DIAG (dev tool; results of DIAG != 0 are wrong by construction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

B, S0, Sq, H, hd = 128, 64, 64, 12, 64
qkv = torch.randn(B * Sq, 3 * H * hd, device="cuda")
kv0 = torch.randn(B * S0, 2 * H * hd, device="cuda")
K.attention_fwd_prefix(kv0, S0, qkv, Sq, B, H, hd, want_lse=True)
t = min(timeit(lambda: K.attention_fwd_prefix(kv0, S0, qkv, Sq, B, H, hd, want_lse=True), 50) for _ in range(5))
print(f"DIAG={os.environ.get('ACT_ATTN_FWD_DIAG', '0'):>3s}  fwd {t*1e3:6.1f} us")
