#!/usr/bin/env python3
"""What would batching the student's weight gradients ACROSS layers buy?  Per layer today: two grouped launches (fc2+fc1 with 7 K ranges, proj+qkv with 14)
on 1,792 token rows.  Here: 8 problems per launch (4 layers of one pair type, or 2 layers of all four products) by K-range count (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

for T, D, Hd in ((1792, 384, 1536), (8192, 384, 1536)):
    mk = lambda M, N: (torch.randn(T, M, device="cuda"), torch.randn(T, N, device="cuda"))
    A = [(D, Hd), (Hd, D)]; Bp = [(D, D), (3 * D, D)]
    fl = lambda dims: sum(2.0 * T * M * N for M, N in dims)
    best = lambda f: min(timeit(f, 30) for _ in range(3))
    pa = [mk(*d) for d in A]; pb = [mk(*d) for d in Bp]
    ta = best(lambda: K.gemm_tn_grouped(pa)); tb = best(lambda: K.gemm_tn_grouped(pb))
    print(f"T={T}: today per layer  A {ta*1e3:6.1f} us ({fl(A)/ta/1e9:5.1f} TF)  B {tb*1e3:6.1f} us ({fl(Bp)/tb/1e9:5.1f} TF)  -> {(ta+tb)*1e3:6.1f} us / layer", flush=True)
    for name, dims, layers in (("4 layers A", A * 4, 4), ("4 layers B", Bp * 4, 4), ("2 layers A+B", (A + Bp) * 2, 2)):
        pairs = [mk(*d) for d in dims]
        line = f"   {name:14s} {fl(dims)/1e9:6.2f} GF:"
        for sp in (1, 2, 3, 4, 7, 14):
            try:
                t = best(lambda: K.gemm_tn_grouped(pairs, splits=sp))
                line += f"  sp{sp}: {t*1e3/layers:6.1f} us/layer ({fl(dims)/t/1e9:5.1f} TF)"
            except Exception as e:
                line += f"  sp{sp}: n/a"
        print(line, flush=True)
