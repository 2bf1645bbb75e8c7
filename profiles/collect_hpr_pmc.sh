#!/bin/bash
# SQ counters of the hull kernels (profiles/hpr_trace_probe.py: 8 keyframes of C3 under PCP_CULL_HPR), one pass per counter
# block, summarised per kernel.  Run through gpurun from the repo root:  bash profiles/collect_hpr_pmc.sh <tag>
set -e -o pipefail
TAG=${1:-r03m_hpr}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
CMD="python3 $R/profiles/hpr_trace_probe.py"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- $CMD > /dev/null 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/${TAG}_p1 -- $CMD > /dev/null 2> $OUT/${TAG}_p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_p2 -- $CMD > /dev/null 2> $OUT/${TAG}_p2.err
cd $R
python3 - $OUT/${TAG}_pmc.json $OUT/${TAG}_trace $OUT/${TAG}_p1 $OUT/${TAG}_p2 <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, trace_dir, passes = sys.argv[1], sys.argv[2], sys.argv[3:]
def short(name):
    name = name.split("(")[0]
    return name.split("::")[-1] if "k_hpr" in name else None
res = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            res[k]["duration_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[k]["duration_us"] += 1
for d in passes:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                res[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
summary = {k: {c: round(v / max(cnt[k][c], 1), 1) for c, v in sorted(cs.items())} | {"dispatches": cnt[k]["duration_us"]} for k, cs in res.items()}
for k, v in summary.items():
    if v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE"):
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        v["valu_busy"] = round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc, 3)
        v["waves_per_simd"] = round(v["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / cyc, 2)
        v["valu_share_of_instructions"] = round(v["SQ_INSTS_VALU"] / (v["SQ_INSTS_VALU"] + v.get("SQ_INSTS_SALU", 0) + v.get("SQ_INSTS_LDS", 0) + v.get("SQ_INSTS_VMEM_RD", 0)), 3)
        v["lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3)
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v[c] for c in ("duration_us", "valu_busy", "waves_per_simd", "valu_share_of_instructions", "lane_utilisation") if c in v} for k, v in summary.items()}, indent=1))
PY
python3 profiles/build_stamp.py $OUT/${TAG}_pmc.json > /dev/null  # the library these counters belong to (bench.py checks it)
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_trace
