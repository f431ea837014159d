#!/bin/bash
# keep a compact kernel timeline (start, end, queue, kernel, grid) of the overlapped production loop for offline analysis
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/gaps.log 2>&1
cd $R
f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with gzip.open("gpurun_out/r05_timeline.csv.gz", "wt") as f:
    for r in rows:
        f.write("%d,%d,%s,%s,%s\n" % (int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Queue_Id"], r["Grid_Size_X"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
print(len(rows))
PY
rm -rf gpurun_out/gaps
