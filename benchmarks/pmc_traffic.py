#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python benchmarks/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Kernel names are folded onto the profiler ids bench.py reports (sgemm_nt / sgemm_nn / sgemm_tn by the operand-layout template
arguments).  Units / corrections: both counters are in KiB (MI355X_MICROARCH.md, HBM section, which also warns that FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950).  Rather than trusting a fixed factor, the script calibrates BOTH counters
in the same run on a streaming kernel of this repo whose byte counts are known exactly: affine_act_kernel (BatchNorm apply)
reads and writes one [262144, C] fp32 tensor per launch, C = 128 and 512 twice each per Stage-II step, i.e. 335,544,320 B
read and written per launch on average; bn_bwd_apply_kernel (reads 2x, writes 1x that) is reported as a cross-check.
Traffic is counted at the L2 <-> fabric boundary, i.e. it includes Infinity-Cache hits."""
import collections
import csv
import glob
import json
import re
import sys


def fold(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    m = re.match(r"sgemm_kernel<\d+, \d+, \d+, (true|false), (true|false)", n)
    if not m:
        m = re.match(r"sgemm16_kernel<\d+, \d+, (true|false), (true|false)", n)
    if m:
        ak, bk = m.group(1) == "true", m.group(2) == "true"
        return "sgemm_nt" if ak and bk else ("sgemm_nn" if ak else ("sgemm_tn" if not bk else "sgemm_tt"))
    if n.startswith("sgemm_nt16_kernel") or n.startswith("sgemm_nt32_kernel") or n.startswith("sgemm_nt_asm_kernel"):
        return "sgemm_nt"
    m = re.match(r"sgemm_q_asm_kernel<\d+, \d+, (true|false)", n)        # round 4: NN / TN on the hand-scheduled loop (A_K = true: NN)
    if m:
        return "sgemm_nn" if m.group(1) == "true" else "sgemm_tn"
    m = re.match(r"sgemm_q16_kernel<\d+, \d+, (true|false), (true|false)", n)
    if m:
        return "sgemm_nn" if m.group(1) == "true" else "sgemm_tn"
    if n.startswith("sgemm_tn_skinny_kernel") or n.startswith("sgemm_tn_grouped_kernel") or n.startswith("sgemm_tn_grouped_asm_kernel"):
        return "sgemm_tn"
    return re.sub(r"<.*", "", n)[:60]


def collect(d, counter):
    tot = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            e = tot[fold(r["Kernel_Name"])]
            e[0] += float(r["Counter_Value"]); e[1].add((f, r["Dispatch_Id"]))
    return {k: (v[0] / max(1, len(v[1])), len(v[1])) for k, v in tot.items()}


def main():
    # round 2: the BatchNorm apply is fused into the consuming GEMM, so the calibration kernel is the BatchNorm backward apply pass
    # (bn_bwd_apply_kernel: reads x and dy, writes dx, each [262144, C] fp32, C = 128 and 512 once per Stage-II step)
    # round 3: per-workload calibration bytes (argv[3] = c2 | s1 | c5): mean [rows, C] fp32 tensor size over the bn_bwd_apply launches of a step --
    # c2: mini-PointNet BN(128), BN(512) on 128*64*32 rows; s1: those two + the FoldingNet BN(512) x 2 on 128*64*32 rows; c5: 32*512*64 rows
    # round 4: the kernel is bn_bwd_apply4_kernel (four columns per thread), and in Stage II the BN(512) launch reads dy only for the rows of the
    # visible patches (13 of 64 groups per cloud at c2, 103 of 512 at c5; act_bn_bwd_groups_f32): read bytes = x (all rows) + dy (live rows)
    WORKLOAD = sys.argv[3] if len(sys.argv) > 3 else "c2"
    CAL_KERNEL = "bn_bwd_apply4_kernel"
    CAL_BYTES = {"c2": 262144.0 * (128 + 512) / 2 * 4, "s1": 262144.0 * (128 + 512 + 512 + 512) / 4 * 4,
                 "c5": 1048576.0 * (128 + 512) / 2 * 4}[WORKLOAD]
    CAL_READ = {"c2": 262144.0 * (2 * 128 + (1 + 13.0 / 64) * 512) / 2 * 4, "s1": 2.0 * CAL_BYTES,
                "c5": 1048576.0 * (2 * 128 + (1 + 103.0 / 512) * 512) / 2 * 4}[WORKLOAD]
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    kf = CAL_READ / (fetch[CAL_KERNEL][0] * 1024.0)
    kw = CAL_BYTES / (write[CAL_KERNEL][0] * 1024.0)
    out = {"workload": WORKLOAD, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-instrument",
           "units": "bytes per launch = counter [KiB] * 1024 * calibration factor",
           "calibration": {"kernel": CAL_KERNEL, "known_bytes_written_per_launch": CAL_BYTES, "known_bytes_read_per_launch": CAL_READ,
                           "fetch_factor": kf, "write_factor": kw,
                           "cross_check": "colsum / layernorm / affine kernels of known size in the same file"},
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fb = fetch.get(k, (0.0, 0))[0] * 1024.0 * kf
        wb = write.get(k, (0.0, 0))[0] * 1024.0 * kw
        out["kernels"][k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                             "launches_profiled": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
