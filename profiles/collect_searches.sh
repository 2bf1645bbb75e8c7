#!/bin/bash
set -e -o pipefail
TAG=${1:-r05_searches}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export PCP_HPR_LANES=1
if [ -n "$2" ]; then export PCP_TILT_BUDGET=$2; fi
python3 $R/profiles/hpr_searches_probe.py 2> $OUT/${TAG}_debug.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/hpr_pass_probe.py > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
cd $R
CSV=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python3 profiles/join_searches.py $OUT/${TAG}_debug.log $CSV $OUT/${TAG}_join.json > /dev/null
grep -v "gave up on" $OUT/${TAG}_debug.log | grep "rows of 16): [0-9]* wavefronts" > $OUT/${TAG}_slots.txt || true
rm -rf $OUT/${TAG}_trace
