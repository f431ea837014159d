#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid size): count, total ms, avg us.  usage: trace_by_grid.py <csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:58]
    grid = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    k = (name, grid)
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
print(f"total kernel ms/step: {tot / steps:.2f}")
for (name, grid), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:58s} grid={'x'.join(grid):18s} n/step={n / steps:6.1f} ms/step={ms / steps:7.3f} avg_us={1e3 * ms / n:8.1f}")
