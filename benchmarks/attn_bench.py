#!/usr/bin/env python3
"""attention forward / backward kernel timings on the shapes of the three workloads (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

for name, (B, S0, Sq, H) in {"stage1 prompt-prefix": (128, 64, 64, 12), "finetune S=65": (32, 0, 65, 6), "student enc S=14": (128, 0, 14, 6), "S=16": (128, 0, 16, 6), "S=7": (32, 0, 7, 2),
                             "student dec S=64": (128, 0, 64, 6), "S=128": (128, 0, 128, 12), "stress S=512": (32, 0, 512, 12),
                             "stress enc S=104": (32, 0, 104, 12), "stress teacher 64+512": (32, 64, 512, 12)}.items():
    hd = 64
    qkv = torch.randn(B * Sq, 3 * H * hd, device="cuda"); do = torch.randn(B * Sq, H * hd, device="cuda")
    kv0 = torch.randn(B * max(S0, 1), 2 * H * hd, device="cuda")
    out, lse = K.attention_fwd_prefix(kv0, S0, qkv, Sq, B, H, hd, want_lse=True)
    fl = 4.0 * B * H * Sq * (S0 + Sq) * hd
    if S0 == 0:                                          # the packed-qkv entry points (S <= 16: register-resident kernels)
        out, lse = K.attention_fwd(qkv, B, Sq, H, hd)
        tf = min(timeit(lambda: K.attention_fwd(qkv, B, Sq, H, hd), 50) for _ in range(3))
        tb = min(timeit(lambda: K.attention_bwd(qkv, out, do, lse, B, Sq, H, hd), 50) for _ in range(3))
    else:
        tf = min(timeit(lambda: K.attention_fwd_prefix(kv0, S0, qkv, Sq, B, H, hd, want_lse=True), 20) for _ in range(3))
        tb = min(timeit(lambda: K.attention_bwd_prefix(kv0, S0, qkv, Sq, out, do, lse, B, H, hd), 20) for _ in range(3))
    print(f"{name:22s} B={B} S0={S0} Sq={Sq} H={H}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:5.1f} TF)  bwd {tb*1e3:7.1f} us ({2.5*fl/tb/1e9:5.1f} TF)")
