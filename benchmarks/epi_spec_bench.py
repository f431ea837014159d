#!/usr/bin/env python3
"""Plain NT GEMM rates of the b128 tiles (10: 128x128, 11: 128x64, 12: 64x64) -- run once with ACT_GEMM_EPI_SPEC=0 (run-time activation switch in the
epilogue: ~95 KB of code per kernel) and once with =1 (kernels instantiated per activation).  Dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

SH = [(8192, 3072, 768), (8192, 2304, 768), (8192, 1536, 768), (8192, 768, 768), (8192, 8192, 2304), (262144, 512, 256), (262144, 256, 128),
      (1792, 1152, 384), (1792, 1536, 384), (8192, 1536, 384)]


def timeit(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for M, N, Kd in SH:
    a = torch.randn(M, Kd, device="cuda"); b = torch.randn(N, Kd, device="cuda"); out = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * N * Kd / 1e9
    reps = max(3, min(30, int(1e3 / fl)))
    line = []
    for tile in (10, 11, 12):
        t = min(timeit(lambda: K.gemm(a, b, True, True, out=out, cfg=(tile, 1)), reps) for _ in range(3))
        t2 = min(timeit(lambda: K.gemm(a, b, True, True, bias=bias, res=res, out=out, cfg=(tile, 1)), reps) for _ in range(3))
        t3 = min(timeit(lambda: K.gemm(a, b, True, True, bias=bias, act=K.EPI_GELU, out=out, cfg=(tile, 1)), reps) for _ in range(3))
        line.append(f"{tile}: plain {fl / t:6.1f} bias+res {fl / t2:6.1f} bias+gelu {fl / t3:6.1f}")
    print(f"{M:6d}x{N:5d}x{Kd:5d}  " + " | ".join(line), flush=True)
