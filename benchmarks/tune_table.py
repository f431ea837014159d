#!/usr/bin/env python3
"""Re-tunes every GEMM shape listed in act_amd/gemm_tune_gfx950.json (or a dump passed as argv[1]) with more repetitions than
the in-step autotuner uses, and writes the refreshed table to argv[2] (default gpurun_out/gemm_tune_gfx950.json).

    ACT_GEMM_TUNE_TABLE=0 python benchmarks/tune_table.py [shapes.json] [out.json]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import act_amd.kernels as K

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "act_amd", "gemm_tune_gfx950.json")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "gemm_tune_gfx950.json")
table = json.load(open(src))
out = {}
dev = torch.device("cuda:0")
ws = K.workspace(dev)
only = os.environ.get("ACT_TUNE_ONLY", "")            # "nt": re-tune the NT (1,1,...) shapes only, "nntn": the others only; the rest of the entries are kept
min_flops = float(os.environ.get("ACT_TUNE_MIN_FLOPS", "0"))
for key in table["configs"]:
    ak, bk, M, N, Kd = (int(v) for v in key.split(","))
    if (only == "nt" and not (ak and bk)) or (only == "nntn" and (ak and bk)) or 2.0 * M * N * Kd < min_flops:
        out[key] = table["configs"][key]
        continue
    a = torch.randn((M, Kd) if ak else (Kd, M), device=dev)
    b = torch.randn((N, Kd) if bk else (Kd, N), device=dev)
    best, t = K.gemm_tune(a, b, ak, bk, M, N, Kd, ws, reps=8, rounds=3)
    out[key] = list(best)
    print(f"{key:28s} -> tile {best[0]:2d} split {best[1]:2d}  {t * 1e3:8.1f} us  {2.0 * M * N * Kd / t / 1e9:6.1f} TF   (was {table['configs'][key]})", flush=True)
    del a, b
table["configs"] = out
table["note"] = "winners of act_amd.kernels.gemm_tune(reps=8, rounds=3) on one MI355X for the GEMM shapes of bench.py --stage 1..4"
json.dump(table, open(dst, "w"), indent=0)
