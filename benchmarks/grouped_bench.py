#!/usr/bin/env python3
"""grouped weight-gradient launch (csrc/gemm_grouped.hip) vs the per-GEMM path on the student's block shapes, by K-range count (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

for T, D, Hd in ((1792, 384, 1536), (8192, 384, 1536), (3296, 768, 3072)):
    groups = {"A fc2_w+fc1_w": [(D, Hd), (Hd, D)], "B proj_w+qkv_w": [(D, D), (3 * D, D)], "all four": [(D, Hd), (Hd, D), (D, D), (3 * D, D)]}
    for name, dims in groups.items():
        pairs = [(torch.randn(T, M, device="cuda"), torch.randn(T, N, device="cuda")) for M, N in dims]
        fl = sum(2.0 * T * M * N for M, N in dims)
        single = 0.0
        for dy, x in pairs:                                  # the shipped per-GEMM configuration + the bias column sum
            single += min(timeit(lambda: (K.gemm(dy, x, False, False), K.colsum(dy)), 30) for _ in range(3))
        line = f"T={T} {name:16s} {fl/1e9:6.2f} GF  per-GEMM+colsum {single*1e3:7.1f} us |"
        for sp in (1, 2, 3, 4, 6, 7, 8, 12, 14, 16):
            try:
                t = min(timeit(lambda: K.gemm_tn_grouped(pairs, splits=sp), 30) for _ in range(3))
                line += f" sp{sp}: {t*1e3:6.1f}"
            except Exception:
                line += f" sp{sp}:   n/a"
        print(line, flush=True)
