#!/usr/bin/env python3
"""one NT GEMM shape with the step's epilogue, a few launches of our kernel and of torch.addmm (hipBLASLt): target of the rocprofv3 --pmc passes in
benchmarks/scripts/gemm_pmc.sh (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
M, N, Kd, tile = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8192, 3072, 768, 11))]
a = torch.randn(M, Kd, device="cuda"); b = torch.randn(N, Kd, device="cuda"); bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(6):
    K.gemm(a, b, True, True, bias=bias, res=res, out=out, cfg=(tile, 1))
    torch.addmm(res, a, b.t(), out=out)
torch.cuda.synchronize()
