#!/usr/bin/env python3
"""Per-shape fp32 GEMM microbenchmark: act_sgemm_f32 (libact_hip.so) vs torch.mm (rocBLAS/hipBLASLt) on the shapes of the
Stage-II step.  hipEvent timing, interleaved rounds.  Dev tool (not part of the product path)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import act_amd.kernels as K

SHAPES = [  # (tag, a_kmajor, b_kmajor, M, N, K)
    ("vit.qkv", 1, 1, 16384, 2304, 768), ("vit.proj", 1, 1, 16384, 768, 768), ("vit.fc1", 1, 1, 16384, 3072, 768),
    ("vit.fc2", 1, 1, 16384, 768, 3072), ("pn.conv2", 1, 1, 262144, 256, 128), ("pn.conv3", 1, 1, 262144, 512, 256),
    ("pn.conv4", 1, 1, 262144, 384, 512), ("dgcnn.l5", 1, 1, 8192, 8192, 2304), ("enc.qkv", 1, 1, 1792, 1152, 384),
    ("enc.fc1", 1, 1, 1792, 1536, 384), ("enc.fc2", 1, 1, 1792, 384, 1536), ("dec.qkv", 1, 1, 8192, 1152, 384),
    ("pn.dA3", 1, 0, 262144, 512, 384), ("pn.dH2", 1, 0, 262144, 256, 512), ("enc.dfc1", 1, 0, 1792, 384, 1536),
    ("pn.dW4", 0, 0, 384, 512, 262144), ("pn.dW3", 0, 0, 512, 256, 262144), ("enc.dWqkv", 0, 0, 1152, 384, 1792),
    ("dec.dWfc1", 0, 0, 1536, 384, 8192),
    # Stage-I backward through the frozen Transformer (dX only) and the tokenizer
    ("s1.dfc2", 1, 0, 8192, 3072, 768), ("s1.dfc1", 1, 0, 8192, 768, 3072), ("s1.dproj", 1, 0, 8192, 768, 768), ("s1.dqkv", 1, 0, 8192, 768, 2304),
    ("s1.dl5", 1, 0, 8192, 2304, 8192), ("s1.dWl5", 0, 0, 8192, 2304, 8192), ("pn.dA1", 1, 0, 262144, 128, 256), ("pn.dW2", 0, 0, 256, 128, 262144),
    ("enc.dWfc2", 0, 0, 384, 1536, 1792), ("enc.dproj", 1, 0, 1792, 384, 384),
    # the frozen teacher Transformer as it runs in the Stage-II step (8,192 patch-token rows / 8,192 prompt rows)
    ("t8.fc1", 1, 1, 8192, 3072, 768), ("t8.fc2", 1, 1, 8192, 768, 3072), ("t8.qkv", 1, 1, 8192, 2304, 768), ("t8.kv", 1, 1, 8192, 1536, 768),
    ("t8.proj", 1, 1, 8192, 768, 768),
]


def timeit(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    only = sys.argv[1:] or None
    res = {}
    for tag, ak, bk, M, N, Kd in SHAPES:
        if only and not any(o in tag for o in only):
            continue
        a = torch.randn((M, Kd) if ak else (Kd, M), device="cuda")
        b = torch.randn((N, Kd) if bk else (Kd, N), device="cuda")
        out = torch.empty(M, N, device="cuda")
        flops = 2.0 * M * N * Kd
        reps = max(3, min(50, int(2e12 / flops)))
        ours = min(timeit(lambda: K.gemm(a, b, bool(ak), bool(bk), out=out), reps) for _ in range(3))
        cfgs = {}
        if os.environ.get("GEMM_BENCH_CFGS"):
            for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18):
                try:
                    tt = min(timeit(lambda: K.gemm(a, b, bool(ak), bool(bk), out=out, cfg=(tile, 1)), reps) for _ in range(2))
                    cfgs[tile] = round(flops / tt / 1e9, 1)
                except Exception:
                    cfgs[tile] = None
        A2 = a if ak else a.t(); B2 = b.t() if bk else b
        ref = min(timeit(lambda: torch.mm(A2, B2, out=out), reps) for _ in range(3))
        res[tag] = dict(M=M, N=N, K=Kd, ours_us=1e3 * ours, ours_tf=flops / ours / 1e9, torch_us=1e3 * ref, torch_tf=flops / ref / 1e9)
        if cfgs:
            print(f"    {tag:10s} per-config TF (1:128x128 2:128x64 3:64x64 4-6: pipelined, 7-9: 16x16x4 MFMA, 10-12: NT b128, 13-14: NN/TN quad): {cfgs}   picked {K._GEMM_CACHE.get((ak, bk, M, N, Kd, 0))}", flush=True)
        print(f"{tag:10s} {M:7d}x{N:5d}x{Kd:6d} ak={ak} bk={bk}  ours {1e3*ours:9.1f} us {flops/ours/1e9:7.1f} TF | torch.mm {1e3*ref:9.1f} us {flops/ref/1e9:7.1f} TF", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
