#!/usr/bin/env python3
"""HBM (fabric) bytes per launch by (kernel, grid size) = per GEMM shape, from the two rocprofv3 PMC passes of benchmarks/pmc_traffic.py.
usage: pmc_traffic_by_grid.py <fetch dir> <write dir> [fetch factor=2.0] [write factor=1.0]   (factors: the calibration of pmc_traffic.py)"""
import collections, csv, glob, sys


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
            grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", "?"))
            e = acc[(name, grid)]
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return acc


ff = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
wf = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
fe, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
rows = []
for k in set(fe) | set(wr):
    f = fe.get(k, [0.0, 1]); w = wr.get(k, [0.0, 1])
    fb = f[0] / max(1, f[1]) * 1024 * ff; wb = w[0] / max(1, w[1]) * 1024 * wf
    rows.append((fb + wb, k, fb, wb, max(f[1], w[1])))
for tot, (name, grid), fb, wb, n in sorted(rows, reverse=True)[:40]:
    print(f"{name:60s} grid={grid:20s} launches={n:4d} fetch={fb/1e6:9.1f} MB write={wb/1e6:9.1f} MB total={tot/1e6:9.1f} MB")
