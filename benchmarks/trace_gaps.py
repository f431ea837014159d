#!/usr/bin/env python3
"""Idle-time analysis of a rocprofv3 kernel_trace.csv of the timed loop (streams overlapping as in production): per HIP queue and for the
union of all queues -- busy time, idle gaps, and the kernels that follow the largest gaps.  usage: trace_gaps.py <csv> [last_n_launches]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
span = (t1 - t0) / 1e6
print(f"launches {len(rows)}  span {span:.2f} ms  sum of kernel durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows) / 1e6:.2f} ms")
# union busy
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union of all queues: busy {busy / 1e6:.2f} ms = {100 * busy / (t1 - t0):.1f} % of the span; idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
hist = collections.Counter(min(int(g / 1e3) // 5 * 5, 50) for g, _ in gaps)
print("  gap histogram (us bucket: count, total ms):", {k: (v, round(sum(g for g, _ in gaps if min(int(g / 1e3) // 5 * 5, 50) == k) / 1e6, 3)) for k, v in sorted(hist.items())})
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    b = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: {len(rs)} launches, busy {b / 1e6:.2f} ms")
# kernels after the largest union gaps
start_of = {int(r["Start_Timestamp"]): r for r in rows}
after = collections.defaultdict(lambda: [0, 0.0])
for g, s in gaps:
    r = start_of[s]
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:50], r["Grid_Size_X"], r["Queue_Id"])
    after[k][0] += 1; after[k][1] += g / 1e3
print("kernels that end an all-idle gap (count, total us):")
for k, (c, us) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k[0]:50s} grid={k[1]:9s} q={k[2]}  n={c:4d}  idle before = {us:8.1f} us  ({us / c:5.1f} each)")
