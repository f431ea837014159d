#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum of counter values / number of dispatches.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]  -> JSON on stdout"""
import collections
import csv
import glob
import json
import sys

out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = name.split("(")[0].replace("void ", "")[:60]
            e = out[key][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"]); e[1].add((f, r["Dispatch_Id"]))
res = {}
for k, cs in out.items():
    res[k] = {c: {"per_launch": v[0] / max(1, len(v[1])), "launches": len(v[1])} for c, v in cs.items()}
print(json.dumps(res, indent=1, sort_keys=True))
