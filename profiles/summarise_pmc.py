#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes -> one JSON (profiles/r01_*).

    python profiles/summarise_pmc.py OUT.json "<command string>" POINTS_PER_LAUNCH DIR [DIR ...]

Every DIR is the -d directory of one `rocprofv3 --pmc ... --output-format csv` pass (counters that do not fit one
pass are collected in separate passes, as MI355X_MICROARCH.md prescribes; never together with a trace).  HBM
traffic of the roofline kernel = 2 x FETCH_SIZE (gfx950 tallies its 128-B read requests at 64 B) + WRITE_SIZE,
both reported in KB by rocprofv3.
"""
import collections
import csv
import glob
import json
import sys


def main():
    out, command, points = sys.argv[1], sys.argv[2], int(sys.argv[3])
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in sys.argv[4:]:
        for fn in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"].split("(")[0]
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    kernels = {}
    for k, cs in sums.items():
        if not k.startswith("pcp::"):
            continue
        kernels[k] = {c + ("_KB_mean" if c.endswith("_SIZE") else "_mean"): v / len(launches[k][c]) for c, v in cs.items()}
        kernels[k]["launches"] = max(len(v) for v in launches[k].values())
    res = {"command": command, "points_per_launch": points, "kernels": kernels}
    pf = kernels.get("pcp::k_project_frame", {})
    if "FETCH_SIZE_KB_mean" in pf and "WRITE_SIZE_KB_mean" in pf:
        fetch = pf["FETCH_SIZE_KB_mean"] * 1024 * 2  # gfx950 correction
        write = pf["WRITE_SIZE_KB_mean"] * 1024
        res["k_project_frame"] = {"fetch_bytes_corrected": fetch, "write_bytes": write,
                                  "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 20 * points}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("k_project_frame", {})))


if __name__ == "__main__":
    main()
