#!/usr/bin/env python3
"""Why does the mini-PointNet conv3 GEMM (262144 x 512 x 256, column statistics fused) take 784 us inside the step and 607 us in an isolated loop?
Isolated launch on ONE buffer set vs rotating over R independent (A, C) sets (every launch touches memory no recent launch touched).  Dev tool."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
import act_amd.composite as CP

R, N, Kd = 262144, 512, 256


def fx_call(a, w, c, fx):
    st = torch.cuda.current_stream().cuda_stream
    epi = K.GemmEpilogue(); epi.alpha = 1.0
    rc = CP.lib.act_sgemm_fx_f32(1, 1, R, N, Kd, a.data_ptr(), Kd, w.data_ptr(), Kd, c.data_ptr(), N, ctypes.byref(epi), ctypes.byref(fx), None, 0, st)
    assert rc == 0, rc


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


w = torch.randn(N, Kd, device="cuda") * 0.05
ts = torch.empty(CP.lib.act_sgemm_fx_tile_stats_floats(R, N), device="cuda")
fx = CP.GemmFx(); fx.tile_stats = ts.data_ptr()
for nset in (1, 2, 4, 8, 16):
    A = [torch.randn(R, Kd, device="cuda") for _ in range(nset)]
    C = [torch.empty(R, N, device="cuda") for _ in range(nset)]
    for i in range(nset):
        fx_call(A[i], w, C[i], fx)
    t = min(timed(lambda i: fx_call(A[i % nset], w, C[i % nset], fx), 2 * max(nset, 8)) for _ in range(3))
    # same, but every launch is preceded by an elementwise pass over ANOTHER 0.8 GB (what the step does between two GEMMs); its time is subtracted
    other = torch.empty(2, R, N // 2 * 2 // 2, device="cuda")
    t_other = min(timed(lambda i: other[0].copy_(other[1]), 16) for _ in range(3))
    t_mix = min(timed(lambda i: (other[0].copy_(other[1]), fx_call(A[i % nset], w, C[i % nset], fx)), 2 * max(nset, 8)) for _ in range(3))
    print(f"buffer sets {nset:2d} ({nset * 0.805:5.1f} GB): {t:7.1f} us per launch = {2.0 * R * N * Kd / t / 1e6:6.1f} TF | after a 0.54 GB copy: {t_mix - t_other:7.1f} us", flush=True)
    del A, C, other
