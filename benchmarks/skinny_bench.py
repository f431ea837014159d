#!/usr/bin/env python3
"""Times the skinny weight-gradient products (TN, one dimension <= 8) of the steps (dev tool; round 4).  ACT_GEMM_SKINNY4=0: the 4-byte-load kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
dev = torch.device("cuda:0")
for M, N, Kd, what in ((128, 3, 262144, "dW of the 3->128 conv, Stage II / I"), (128, 3, 1048576, "the same at C5"), (128, 3, 8192, "position-embedding dW"),
                       (512, 5, 262144, "FoldingNet 5->512"), (3, 512, 262144, "FoldingNet 512->3")):
    a = torch.randn(Kd, M, device=dev); b = torch.randn(Kd, N, device=dev)
    for _ in range(3):
        K.gemm(a, b, False, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.gemm(a, b, False, False)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f"{M:4d} x {N:4d} x {Kd:8d}  {us:8.1f} us  {4.0 * Kd * (M + N) / us / 1e6:6.2f} TB/s   {what}", flush=True)
