#!/usr/bin/env python3
"""DGCNN edge-conv tail (gather + GroupNorm + LeakyReLU + max) forward / backward timings at the Stage-I shapes (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
from gemm_bench import timeit
B, G, k = 128, 64, 4
idx = torch.stack([torch.stack([torch.randint(0, G, (G,)) for _ in range(k)]) for _ in range(B)]).cuda()
for C in (256, 512, 1024):
    gn = torch.nn.GroupNorm(4, C).cuda()
    yz = torch.randn(B * G, 2 * C, device="cuda", requires_grad=True)
    do = torch.randn(B * G, C, device="cuda")
    out = K.edge_gn_lrelu_max_train(yz, C, idx, B, G, k, C, gn)
    tf = min(timeit(lambda: K.edge_gn_lrelu_max_train(yz, C, idx, B, G, k, C, gn), 20) for _ in range(3))
    def fb():
        o = K.edge_gn_lrelu_max_train(yz, C, idx, B, G, k, C, gn); o.backward(do)
    tb = min(timeit(fb, 20) for _ in range(3))
    print(f"edge layer C={C}: fwd {tf*1e3:7.1f} us   fwd+bwd {tb*1e3:7.1f} us")
for C in (8192, 384):
    gn = torch.nn.GroupNorm(4, C).cuda()
    h = torch.randn(B * G, C, device="cuda", requires_grad=True); do = torch.randn(B * G, C, device="cuda")
    tf = min(timeit(lambda: K.edge_gn_lrelu_max_train(h, -1, None, B, G, 1, C, gn), 20) for _ in range(3))
    def fb():
        o = K.edge_gn_lrelu_max_train(h, -1, None, B, G, 1, C, gn); o.backward(do)
    tb = min(timeit(fb, 20) for _ in range(3))
    print(f"head C={C}: fwd {tf*1e3:7.1f} us   fwd+bwd {tb*1e3:7.1f} us")
