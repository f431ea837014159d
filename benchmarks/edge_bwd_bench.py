#!/usr/bin/env python3
"""backward of the DGCNN edge-conv tail (gather + GroupNorm + LeakyReLU + max over k) at the Stage-I shapes: LDS-resident passes (round 6) vs the global-gather /
scatter-image kernels, via the runtime switch act_edge_bwd_lds (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit

B, G, k = 128, 64, 4
for C in (512, 1024):
    torch.manual_seed(0)
    yz = torch.randn(B * G, 2 * C, device="cuda"); do = torch.randn(B * G, C, device="cuda")
    idx = torch.randint(0, G, (B, k, G), device="cuda"); idx[:, 0] = torch.arange(G, device="cuda")
    gn = torch.nn.GroupNorm(4, C).cuda()
    yg = yz.clone().requires_grad_(True)
    out = K.edge_gn_lrelu_max_train(yg, C, idx, B, G, k, C, gn)
    nbytes = 4.0 * B * G * C * 5
    for on in (0, 1, 0, 1):
        K.lib.act_edge_bwd_lds(on)
        t = min(timeit(lambda: torch.autograd.grad(out, yg, do, retain_graph=True), 20) for _ in range(3))
        print(f"C={C} act_edge_bwd_lds={on}: backward (3 launches + 2 column sums) {t*1e3:7.1f} us  ({nbytes/t/1e9:6.0f} GB/s compulsory)")
K.lib.act_edge_bwd_lds(1)
