#!/usr/bin/env python3
"""What the GELU epilogue costs on the fc1 launches (table-selected tiles): bias only vs bias + GELU, and dX with / without the GELU-gradient epilogue.  Dev tool.
On gfx950 a VALU instruction never overlaps an f32 MFMA of the same SIMD (profiles/r06_mfma_valu_kinds.txt): the epilogue's erff is matrix-pipe time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

for name, (M, N, Kd) in {"teacher fc1": (8192, 3072, 768), "student fc1": (1792, 1536, 384), "decoder fc1": (8192, 1536, 384), "C5 student fc1": (3328, 3072, 768)}.items():
    a = torch.randn(M, Kd, device="cuda") * 0.5; w = torch.randn(N, Kd, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    fl = 2.0 * M * N * Kd / 1e9
    t1 = min(timeit(lambda: K.gemm(a, w, True, True, bias=bias, out=out), 30) for _ in range(5))
    t2 = min(timeit(lambda: K.gemm(a, w, True, True, bias=bias, act=K.EPI_GELU, out=out), 30) for _ in range(5))
    print(f"{name:16s} {M}x{N}x{Kd}: bias {t1*1e3:7.1f} us ({fl/t1:6.1f} TF)   bias+gelu {t2*1e3:7.1f} us ({fl/t2:6.1f} TF)   gelu costs {100*(t2-t1)/t1:5.1f} %")
