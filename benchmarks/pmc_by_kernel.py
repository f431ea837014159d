#!/usr/bin/env python3
"""mean counter value per kernel (name filter: a regular expression) over the dispatches of rocprofv3 --pmc output dirs.  usage: pmc_by_kernel.py <filter> <dir>..."""
import collections, csv, glob, re, sys
flt = re.compile(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if flt.search(n):
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"    {c:40s} {sum(v) / len(v):16.1f}   (n={len(v)})")
