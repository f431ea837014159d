#!/usr/bin/env python3
"""Times the two sparse max-pool-backward products (csrc/pool_bwd.hip) at the Stage-II and C5 encoder geometries (dev tool; round 4).
    python benchmarks/pool_bwd_bench.py            (kernel variants are chosen by ACT_POOL_BWD_DX4 / ACT_POOL_BWD_DW2, read once per process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

dev = torch.device("cuda:0")
lib = K.lib
st = torch.cuda.current_stream().cuda_stream
for name, G, n, C in (("stage2", 8192, 32, 384), ("c5", 16384, 64, 768)):
    N = 512
    g = torch.Generator().manual_seed(1)
    dout = torch.randn(G, C, generator=g).to(dev)
    arg = torch.randint(0, n, (G, C), generator=g, dtype=torch.int32).to(dev)
    pat = os.environ.get("POOL_BENCH_ARG", "random")                      # LDS access experiments: every lane its own row / all lanes one row
    if pat == "distinct":
        arg = (torch.arange(C, dtype=torch.int32) % n).repeat(G, 1).to(dev)
    elif pat.startswith("share"):                                         # runs of k consecutive channels on one row
        k = int(pat[5:])
        arg = ((torch.arange(C, dtype=torch.int32) // k) % n).repeat(G, 1).to(dev)
    elif pat == "same":
        arg = torch.zeros(G, C, dtype=torch.int32, device=dev)
    keep = float(os.environ.get("POOL_BENCH_LIVE", "1.0"))               # fraction of groups with a non-zero gradient (Stage II: 13 / 64 visible)
    if keep < 1.0:
        dout[torch.rand(G, generator=g).to(dev) >= keep] = 0
    W = (torch.randn(C, N, generator=g) * 0.1).to(dev)
    X = torch.randn(G * n, N, device=dev)
    sc = torch.rand(N, device=dev) + 0.5; sh = torch.randn(N, device=dev) * 0.2
    dx = torch.empty(G * n, N, device=dev); dw = torch.empty(C, N, device=dev)
    ws = torch.empty((160 << 20) // 4, device=dev)

    def run_dx():
        assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, n, C, W.data_ptr(), N, N, dx.data_ptr(), N, st) == 0

    def run_dw():
        assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, n, C, X.data_ptr(), N, N, sc.data_ptr(), sh.data_ptr(), dw.data_ptr(), N,
                                               ws.data_ptr(), ws.numel() * 4, st) == 0
    for label, fn, nbytes in (("dx", run_dx, 4.0 * G * n * N), ("dw", run_dw, 4.0 * G * n * N)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        print(f"{name} {label}: {us:8.1f} us   dense-operand bytes {nbytes / 1e6:.0f} MB -> {nbytes / us / 1e6:.2f} TB/s", flush=True)
