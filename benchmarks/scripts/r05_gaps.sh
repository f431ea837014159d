#!/bin/bash
# idle-time analysis of the production (overlapped) timed loop: rocprofv3 kernel trace of bench.py + benchmarks/trace_gaps.py
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/gaps.log 2>&1
cd $R
f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
python benchmarks/trace_gaps.py $f > gpurun_out/r05_c2_overlap_gaps.txt
python - "$f" <<'PY' >> gpurun_out/r05_c2_overlap_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
print("\nper-queue gaps >= 20 us (the kernel that ENDS the gap, i.e. what the queue was waiting to start):")
for q, rs in sorted(byq.items()):
    gaps = collections.defaultdict(lambda: [0, 0.0]); tot = 0.0
    for a, b in zip(rs, rs[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if g >= 20:
            k = (b["Kernel_Name"].split("(")[0].replace("void ", "")[:60], b["Grid_Size_X"]); gaps[k][0] += 1; gaps[k][1] += g; tot += g
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    small = span - busy - tot
    print(f"queue {q}: span {span/1e3:.2f} ms, busy {busy/1e3:.2f} ms, gaps >= 20 us {tot/1e3:.2f} ms, gaps < 20 us {small/1e3:.2f} ms over {len(rs)} launches")
    for k, (c, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"    {k[0]:60s} grid={k[1]:9s} n={c:3d} total {us/1e3:7.2f} ms ({us/c:7.1f} us each)")
PY
rm -rf gpurun_out/gaps
cat gpurun_out/r05_c2_overlap_gaps.txt
