#!/usr/bin/env python3
"""The software-pipelined NT tiles (17: 128x128, 18: 128x64 -- fragment double buffering, global prefetch distance 2, bare s_barrier) against 10 / 11 with the
epilogues of the step.  Dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

SH = [(8192, 3072, 768), (8192, 2304, 768), (8192, 1536, 768), (8192, 768, 768), (8192, 768, 3072), (8192, 8192, 2304), (262144, 384, 512)]


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
    for tile, sp in ((10, 1), (17, 1), (11, 1), (18, 1), (10, 2), (17, 2)):
        try:
            t = min(timeit(lambda: K.gemm(a, b, True, True, bias=bias, res=res, out=out, cfg=(tile, sp)), reps) for _ in range(3))
        except Exception:
            line.append(f"{tile}/{sp}: n/a"); continue
        line.append(f"{tile}/{sp}: {fl / t:6.1f}")
    print(f"{M:6d}x{N:5d}x{Kd:5d}  " + " | ".join(line), flush=True)
