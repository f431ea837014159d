#!/usr/bin/env python3
"""NT b128 kernel on explicit tiles 10/11/12 for the teacher / tokenizer shapes (dev A/B tool; compare runs with different
ACT_NT16_* environment knobs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit
for (M, N, Kd) in [(8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (8192, 1536, 768), (8192, 8192, 2304),
                   (262144, 512, 512), (262144, 256, 128), (1792, 1536, 384)]:
    a = torch.randn(M, Kd, device="cuda"); b = torch.randn(N, Kd, device="cuda"); out = torch.empty(M, N, device="cuda")
    ref = None
    line = f"{M}x{N}x{Kd}:"
    for tile in (10, 11, 12):
        c = K.gemm(a, b, True, True, cfg=(tile, 1))
        if ref is None:
            ref = (a[:256].double() @ b.double().t())
        err = ((c[:256].double() - ref).abs().max() / ref.abs().max()).item()
        t = min(timeit(lambda: K.gemm(a, b, True, True, out=out, cfg=(tile, 1)), 20) for _ in range(3))
        line += f"  t{tile} {t*1e3:7.1f}us {2.0*M*N*Kd/t/1e9:6.1f}TF" + ("" if err < 2e-5 else f" ERR{err:.1e}")
    print(line)
