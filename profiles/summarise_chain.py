#!/usr/bin/env python3
"""Per-kernel sums of the counter passes collect_chain.sh writes (csv per pass): for each kernel name the mean per
dispatch of every counter, over the dispatches of the chain's kernels; durations from the kernel trace.
python profiles/summarise_chain.py out.json trace_dir pass_dir..."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, trace_dir, passes = sys.argv[1], sys.argv[2], sys.argv[3:]
KEEP = ("k_sor_select", "k_sor_mean_distance", "k_mls_fit", "k_grid_count", "k_grid_order", "k_grid_scatter", "k_grid_probe", "k_bbox")


def short(name):
    if "k_mls_fit" in name and "<false, true>" in name:  # the tile form, apart from the gather form that takes its leftovers
        return "k_mls_fit_tile"
    for k in KEEP:
        if k in name:
            return k
    return None


res = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            res[k]["duration_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[k]["duration_us"] += 1
for d in passes:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                res[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
summary = {k: {c: round(v / max(cnt[k][c], 1), 1) for c, v in sorted(cs.items())} | {"dispatches": cnt[k]["duration_us"]} for k, cs in res.items()}
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
