#!/usr/bin/env python3
"""Isolated cost of the fused producer / consumer passes of the mini-PointNet GEMMs (act_sgemm_fx_f32) against the plain GEMM of
the same shape plus the separate passes they replace.  hipEvent timing, idle GPU.  Dev tool (not part of the product path)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K
import act_amd.composite as CP


def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def fx_call(M, N, Kd, a, w, c, fx):
    st = torch.cuda.current_stream().cuda_stream
    epi = K.GemmEpilogue(); epi.alpha = 1.0
    rc = CP.lib.act_sgemm_fx_f32(1, 1, M, N, Kd, a.data_ptr(), Kd, w.data_ptr(), Kd, c.data_ptr() if c is not None else None, N,
                                 ctypes.byref(epi), ctypes.byref(fx), None, 0, st)
    assert rc == 0, rc


def main():
    R, n = 262144, 32
    res = {}
    for tag, N, Kd, kind in (("conv2 128->256 affineA+groupmax+store", 256, 128, "ag"), ("conv3 256->512 colstats", 512, 256, "cs"),
                             ("conv4 512->384 affineA+groupmax nostore", 384, 512, "agn")):
        a = torch.randn(R, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * 0.05; c = torch.empty(R, N, device="cuda")
        sc = torch.rand(Kd, device="cuda") + 0.5; sh = torch.randn(Kd, device="cuda") * 0.1
        gm = torch.empty(R // n, N, device="cuda"); ga = torch.empty(R // n, N, device="cuda", dtype=torch.int32)
        ts = torch.empty(CP.lib.act_sgemm_fx_tile_stats_floats(R, N), device="cuda")
        flops = 2.0 * R * N * Kd
        plain = timeit(lambda: K.gemm(a, w, True, True, out=c))
        fx = CP.GemmFx()
        if kind == "cs":
            fx.tile_stats = ts.data_ptr()
            t = timeit(lambda: fx_call(R, N, Kd, a, w, c, fx))
        else:
            fx.a_scale = sc.data_ptr(); fx.a_shift = sh.data_ptr(); fx.gmax = gm.data_ptr(); fx.garg = ga.data_ptr(); fx.group = n
            fx.store_c = 1 if kind == "ag" else 0
            t = timeit(lambda: fx_call(R, N, Kd, a, w, c if kind == "ag" else None, fx))
        res[tag] = {"plain_ms": round(plain, 4), "plain_tf": round(flops / plain / 1e9, 1), "fx_ms": round(t, 4), "fx_tf": round(flops / t / 1e9, 1)}
        print(tag, res[tag], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/fx_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
