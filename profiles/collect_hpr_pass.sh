#!/bin/bash
# Kernel trace of the whole-run hull pass of C3 (profiles/hpr_pass_probe.py) with ONE keyframe in flight: per-kernel totals and
# launch-duration percentiles (profiles/overlap.py).  bash profiles/collect_hpr_pass.sh <tag> [PCP_TILT_BUDGET]
set -e -o pipefail
TAG=${1:-r05_hpr_pass}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export PCP_HPR_LANES=${PCP_HPR_LANES:-1}
if [ -n "$2" ]; then export PCP_TILT_BUDGET=$2; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/hpr_pass_probe.py > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
cd $R
CSV=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
{ echo "# lanes $PCP_HPR_LANES budget ${PCP_TILT_BUDGET:-default} lib $(python3 profiles/build_stamp.py)"; python3 profiles/overlap.py $CSV 0.5; } > $OUT/${TAG}_kernels.txt
rm -rf $OUT/${TAG}_trace
