#!/usr/bin/env python3
"""Every (tile, split-K) candidate of the autotuner on the student's GEMM shapes (1,792 = 128 x 14 encoder rows, 8,192 = 128 x 64 decoder
rows) next to torch.mm (hipBLASLt): where the launch-level parallelism problem of VERDICT r2 weak #5 sits.  Dev tool."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

SH = []
for T in (1792, 8192):
    for (N, Kd) in ((1152, 384), (384, 384), (1536, 384), (384, 1536)):
        SH.append(("nt", 1, 1, T, N, Kd))
    for (N, Kd) in ((1536, 384), (384, 1536), (384, 384), (384, 1152)):
        SH.append(("nn", 1, 0, T, N, Kd))
    for (M, N) in ((384, 1536), (1536, 384), (384, 384), (1152, 384)):
        SH.append(("tn", 0, 0, M, N, T))


def timeit(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = []
for tag, ak, bk, M, N, Kd in SH:
    a = torch.randn((M, Kd) if ak else (Kd, M), device="cuda")
    b = torch.randn((N, Kd) if bk else (Kd, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * Kd
    tr = []
    best, bt = K.gemm_tune(a, b, ak, bk, M, N, Kd, K.workspace(a.device), reps=20, rounds=3, trace=tr)
    A2 = a if ak else a.t(); B2 = b.t() if bk else b
    ref = min(timeit(lambda: torch.mm(A2, B2, out=out), 50) for _ in range(3))
    tr.sort(key=lambda x: x[2])
    top = [(t, s, round(1e3 * ms, 1), round(fl / ms / 1e9, 1)) for t, s, ms in tr[:5]]
    print(f"{tag} {M:5d}x{N:5d}x{Kd:5d}  best {best} {1e3*bt:7.1f} us {fl/bt/1e9:6.1f} TF | torch.mm {1e3*ref:7.1f} us {fl/ref/1e9:6.1f} TF | ideal {fl/157.3e6:6.1f} us | top5 {top}", flush=True)
    res.append(dict(tag=tag, M=M, N=N, K=Kd, best=best, best_us=1e3 * bt, torch_us=1e3 * ref, top=top))
print(json.dumps(res))
