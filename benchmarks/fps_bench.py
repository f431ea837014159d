#!/usr/bin/env python3
"""FPS / kNN-group kernel times on the shapes of the workloads (hipEvents per launch through the library's profiler; dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from act_amd.pointnet2_ops import pointnet2_utils as pu
from act_amd.knn_cuda import knn_group
import act_amd._C as C

for name, (B, N, G, M) in {"C2 128x1024 -> 64 x 32": (128, 1024, 64, 32), "C5 32x8192 -> 512 x 64": (32, 8192, 512, 64), "finetune pool 32x8192 -> 1200": (32, 8192, 1200, 0),
                           "inference 128x8192 -> 1024": (128, 8192, 1024, 0), "finetune 32x1024 -> 64 x 32": (32, 1024, 64, 32), "finetune 8k 32x8192 -> 512 x 32": (32, 8192, 512, 32)}.items():
    x = torch.randn(B, N, 3, device="cuda")
    for _ in range(3):
        i, c = pu.furthest_point_sample_with_centers(x, G)
        if M: knn_group(x, c, M, want_nbr=True)
    C.prof_reset(); C.prof_enable(True)
    for _ in range(10):
        i, c = pu.furthest_point_sample_with_centers(x, G)
        if M: knn_group(x, c, M, want_nbr=True)
    torch.cuda.synchronize(); C.prof_enable(False)
    t = C.prof_table()
    print(f"{name:34s}", {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in t.items()}, "us", flush=True)
