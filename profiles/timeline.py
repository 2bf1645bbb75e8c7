#!/usr/bin/env python3
"""Prints the dispatch timeline (start, duration, gap to the previous end) of the last steps of a rocprofv3
--kernel-trace csv: python profiles/timeline.py <kernel_trace.csv> [n_last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = int(rows[-n]["Start_Timestamp"])
prev_end = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:9.1f}"
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {gap:>9}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0]}")
    prev_end = e
