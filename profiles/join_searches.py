#!/usr/bin/env python3
"""Joins the per-keyframe search counts (hpr_searches_probe.py's stderr) with the launch durations of a 1-lane kernel trace of
hpr_pass_probe.py: duration of k_hpr_tilt<.., 16> by keyframe against its number of searches.
python3 profiles/join_searches.py debug.log kernel_trace.csv out.json"""
import csv, json, re, sys
import numpy as np
log, trace, out = sys.argv[1:4]
searched, cands = [], []
for line in open(log):
    m = re.match(r"hpr: (\d+) candidates, (\d+) searched", line)
    if m:
        cands.append(int(m.group(1))); searched.append(int(m.group(2)))
rows = [r for r in csv.DictReader(open(trace))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
res = {}
for name in ("k_hpr_tilt<false, 16>", "k_hpr_tilt<false, 64>", "k_hpr_radial", "k_hpr_quick"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    d = np.array(d[-256:])
    if len(d) != 256 or len(searched) != 256:
        res[name] = {"launches": len(d), "keyframes_logged": len(searched)}
        continue
    s = np.array(searched if "tilt" in name else cands, float)
    A = np.vstack([s, np.ones_like(s)]).T
    coef, *_ = np.linalg.lstsq(A, d, rcond=None)
    fit = A @ coef
    res[name] = {"us_per_1000_items": round(float(coef[0]) * 1e3, 2), "intercept_us": round(float(coef[1]), 1),
                 "corr": round(float(np.corrcoef(s, d)[0, 1]), 4), "total_ms": round(float(d.sum()) / 1e3, 2),
                 "residual_ms": round(float(np.abs(d - fit).sum()) / 1e3, 2),
                 "worst": [{"keyframe": int(k), "items": int(s[k]), "us": round(float(d[k]), 1), "fit_us": round(float(fit[k]), 1)}
                           for k in np.argsort(-(d - fit))[:8]]}
res["searched_total"] = int(sum(searched)); res["candidates_total"] = int(sum(cands))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
