#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes -> one JSON (profiles/rNN_pmc.json), read back by bench.py.

    python profiles/summarise_pmc.py OUT.json "<command string>" STEP_POINTS STEP_KEYFRAMES ROOFLINE_POINTS DIR [DIR ...]

Every DIR is the -d directory of one `rocprofv3 --pmc ... --output-format csv` pass of the SAME bench.py command
(counters that do not fit one pass are collected in separate passes, as MI355X_MICROARCH.md prescribes; never together
with a trace).  HBM traffic = 2 x FETCH_SIZE (gfx950 tallies its 128-B read requests at 64 B) + WRITE_SIZE, both
reported in KB by rocprofv3.

k_project_frame is launched on two clouds by bench.py: the workload's own (STEP_POINTS, cache resident at 10 M points)
and the roofline cloud (ROOFLINE_POINTS, outside the Infinity Cache); the dispatches are told apart by their grid
size (one lane per 4 points, 256-thread workgroups).
"""
import collections
import csv
import glob
import json
import re
import sys


def grid_of(points):
    quads = (points + 3) // 4
    return ((quads + 255) // 256) * 256


def main():
    out, command = sys.argv[1], sys.argv[2]
    step_points, keyframes, roof_points = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in sys.argv[6:]:
        for fn in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                # "void pcp::k_depth_pass<true>(float const*, ...)" -> "pcp::k_depth_pass" (the batched kernels are templates
                # since round 3: the specialisations of one kernel are one row)
                k = re.sub(r"<.*>$", "", re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]))
                if k == "pcp::k_project_frame":
                    g = int(r["Grid_Size"])
                    k += "@hbm" if g == grid_of(roof_points) else ("@ic" if g == grid_of(step_points) else f"@grid{g}")
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add((fn, r["Dispatch_Id"]))
    kernels = {}
    for k, cs in sums.items():
        if not k.startswith("pcp::"):
            continue
        kernels[k] = {c + ("_KB_mean" if c.endswith("_SIZE") else "_mean"): v / len(launches[k][c]) for c, v in cs.items()}
        kernels[k]["launches"] = max(len(v) for v in launches[k].values())
    res = {"command": command, "step": {"points": step_points, "keyframes": keyframes, "kernels": kernels}}
    for tag, pts, key in (("@hbm", roof_points, "k_project_frame_hbm"), ("@ic", step_points, "k_project_frame_ic_resident")):
        pf = kernels.get("pcp::k_project_frame" + tag, {})
        if "FETCH_SIZE_KB_mean" in pf and "WRITE_SIZE_KB_mean" in pf:
            fetch = pf["FETCH_SIZE_KB_mean"] * 1024 * 2  # gfx950 correction
            write = pf["WRITE_SIZE_KB_mean"] * 1024
            res[key] = {"points_per_launch": pts, "fetch_bytes_corrected": fetch, "write_bytes": write,
                        "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 20 * pts,
                        "launches": pf["launches"]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in res if k.startswith("k_project")}))


if __name__ == "__main__":
    main()
