#!/usr/bin/env python3
"""Overlap summary of a rocprofv3 --kernel-trace csv, last `frac` of the run: span, union of busy time, sum of kernel
durations, per-kernel totals, per-queue busy time.   python profiles/overlap.py <kernel_trace.csv> [start_fraction]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [r for r in rows if "k_hpr" in r["Kernel_Name"] or "scan" in r["Kernel_Name"]]
rows = rows[int(len(rows) * frac):]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
span = max(e for _, e in iv) - min(s for s, _ in iv)
tot = sum(e - s for s, e in iv)
union, cur_s, cur_e = 0, None, None
for s, e in sorted(iv):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
per = defaultdict(lambda: [0, 0])
durs = defaultdict(list)
perq = defaultdict(int)
for r, (s, e) in zip(rows, iv):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    per[k][0] += e - s
    per[k][1] += 1
    durs[k].append((e - s) / 1e3)
    perq[r.get("Queue_Id", "?")] += e - s
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy (union) {union / 1e6:.2f} ms  sum of durations {tot / 1e6:.2f} ms  "
      f"idle {100 * (1 - union / span):.1f} %  mean concurrency while busy {tot / union:.2f}")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    d = sorted(durs[k])
    print(f"  {k:28s} {t / 1e6:9.2f} ms  {c:6d} launches  {t / c / 1e3:8.1f} us each   min {d[0]:.1f}  p10 {d[len(d) // 10]:.1f}  median {d[len(d) // 2]:.1f}  "
          f"p90 {d[(9 * len(d)) // 10]:.1f}  max {d[-1]:.1f}")
for q, t in sorted(perq.items()):
    print(f"  queue {q}: {t / 1e6:.2f} ms")
