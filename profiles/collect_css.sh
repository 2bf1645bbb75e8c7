#!/bin/bash
# kernel stats of the streamed chain on the whole C3 map (profiles/css_probe.py)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_css}
REPS=${2:-2}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/css_probe.py 28 $REPS > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# the launches of the heavy kernels in order (start, duration): which sweep costs what
python3 - $OUT/${TAG}_trace $OUT/${TAG}_heavy_launches.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("k_voxel_emit", "k_sor_select", "k_sor_wave", "k_brick_expand"):
            if key in n:
                rows.append((int(r["Start_Timestamp"]), key, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(sys.argv[2], "w") as o:
    for t, k, ms, g in rows:
        o.write(f"{(t - t0) / 1e6:10.1f} ms  {k:16s} {ms:8.2f} ms  grid {g}\n")
PY
rm -rf $OUT/${TAG}_trace
