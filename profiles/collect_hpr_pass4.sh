#!/bin/bash
# Kernel trace of the whole-run hull pass of C3 with the default keyframes in flight: union of busy time, concurrency, per-kernel
# totals under overlap, per-queue busy time (profiles/overlap.py).  bash profiles/collect_hpr_pass4.sh <tag> [lanes]
set -e -o pipefail
TAG=${1:-r05_hpr_pass4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
if [ -n "$2" ]; then export PCP_HPR_LANES=$2; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/hpr_pass_probe.py > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
cd $R
CSV=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
{ echo "# lanes ${PCP_HPR_LANES:-default} lib $(python3 profiles/build_stamp.py)"; cat $OUT/${TAG}_probe.json; python3 profiles/overlap.py $CSV 0.5; } > $OUT/${TAG}_kernels.txt
rm -rf $OUT/${TAG}_trace
