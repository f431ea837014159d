#!/usr/bin/env python3
"""Which Tensile kernels does torch.mm (hipBLASLt) pick for the teacher's fp32 products?  Run under rocprofv3 --kernel-trace.  Dev tool."""
import torch
for M, N, K in ((8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (8192, 2304, 768), (8192, 8192, 2304), (1792, 384, 384), (1792, 1536, 384)):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        torch.mm(a, b.t(), out=out)
    torch.cuda.synchronize()
