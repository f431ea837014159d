#!/usr/bin/env python3
"""Per-kernel MfmaUtil (rocprofv3 derived counter: MFMA-busy cycles / (GPU-active cycles * SIMDs), percent) from one PMC pass:

    ACT_OVERLAP_TEACHER=0 ACT_OVERLAP_DW=0 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d gpurun_out/pmc_mfma -- \\
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument
    python benchmarks/pmc_mfma_util.py gpurun_out/pmc_mfma > profiles/rNN_pmc_mfma_util.json
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import fold                     # kernel-name folding shared with the traffic script
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])
raw = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])      # the same per kernel FUNCTION (template arguments stripped): compiler- vs hand-scheduled GEMM kernels apart
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "MfmaUtil":
            continue
        e = acc[fold(r["Kernel_Name"])]
        d = dur.get(r["Dispatch_Id"], 0)
        e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += float(r["Counter_Value"]) * d; e[3] += d
        q = raw[r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]]
        q[0] += float(r["Counter_Value"]); q[1] += 1; q[2] += float(r["Counter_Value"]) * d; q[3] += d
out = {"source": "rocprofv3 --kernel-trace --pmc MfmaUtil (own pass, auxiliary streams serialised) -- python bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-instrument",
       "unit": "percent of MFMA-pipe busy cycles; duration-weighted mean (and plain mean) over launches",
       "kernels": {k: {"mfma_util_percent_time_weighted": (v[2] / v[3] if v[3] else None), "mfma_util_percent_mean": v[0] / v[1],
                       "launches_profiled": v[1]} for k, v in sorted(acc.items()) if v[0] > 0},
       "gemm_kernel_functions": {k: {"mfma_util_percent_time_weighted": (v[2] / v[3] if v[3] else None), "launches_profiled": v[1],
                                     "ms_profiled": v[3] / 1e6} for k, v in sorted(raw.items()) if v[0] > 0 and k.startswith("sgemm")}}
print(json.dumps(out, indent=1))
