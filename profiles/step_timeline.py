#!/usr/bin/env python3
"""Dispatch timeline of whole steps inside a rocprofv3 --kernel-trace csv of bench.py: start, duration, gap to the previous
end, queue.  python3 profiles/step_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_colour_pass" in r["Kernel_Name"]]
mid = idx[len(idx) // 2]
lo = max(0, mid - 16)
prev_end = None
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:mid + 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else "%8.1f" % ((s - prev_end) / 1e3)
    name = r["Kernel_Name"].split("(")[0][:60]
    print("%9.1f us dur %8.1f gap %8s q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), name))
    prev_end = e
