#!/bin/bash
# every launch of one enableMLS chain (SOR -> MLS -> SOR, 10 M points, upsampling NONE) in order: start, duration, the gap in front
# (profiles/chain_probe.py; the last of its calls)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_chain_tl}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/chain_probe.py 10000000 chain-only > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
python3 - $OUT/${TAG}_trace $OUT/${TAG}.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcp::", "")[:44]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
rows.sort()
# the last chain: from the last k_bbox-free gap of > 20 ms backwards is fragile; print the last 140 launches
rows = rows[-140:]
t0 = rows[0][0]
prev = t0
with open(sys.argv[2], "w") as o:
    for s, e, n in rows:
        o.write(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  {(e - s) / 1e3:8.1f} us  {n}\n")
        prev = e
PY
rm -rf $OUT/${TAG}_trace
