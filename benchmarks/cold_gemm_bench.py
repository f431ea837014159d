#!/usr/bin/env python3
"""Does the in-step GEMM rate (fc1 119 TF in the step trace vs 129 TF in a hot loop) come from operands that are cold in L2 / Infinity Cache?
Each launch is timed alone (hipEvents) after  (a) nothing: hot loop,  (b) a 2 GB sweep that evicts L2 + MALL,  (c) sweep, then A touched (as if the
producer kernel had just written it), W cold,  (d) sweep, then A and W touched (W prefetched).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

SHAPES = [("t8.fc1", 8192, 3072, 768), ("t8.fc2", 8192, 768, 3072), ("t8.qkv", 8192, 2304, 768), ("t8.kv", 8192, 1536, 768), ("t8.proj", 8192, 768, 768),
          ("pn.conv3", 262144, 512, 256), ("dgcnn.l5", 8192, 8192, 2304), ("enc.fc1", 1792, 1536, 384)]
big = torch.empty(512 << 20, dtype=torch.float32, device="cuda")      # 2 GB
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def once(fn, prep):
    prep(); torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for tag, M, N, Kd in SHAPES:
    a = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda"); out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * Kd
    fn = lambda: K.gemm(a, w, True, True, out=out)
    for _ in range(3): fn()
    modes = {"hot": lambda: None, "all cold": lambda: big.fill_(1.0), "A hot, W cold": lambda: (big.fill_(1.0), a.mul_(1.0)),
             "A hot, W prefetched": lambda: (big.fill_(1.0), a.mul_(1.0), w.mul_(1.0))}
    line = f"{tag:9s} {M}x{N}x{Kd}:"
    for name, prep in modes.items():
        t = sorted(once(fn, prep) for _ in range(7))[3]
        line += f"  {name} {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF |"
    print(line, flush=True)
