#!/bin/bash
# the launches of a few whole steps of bench.py in order, with the gap in front of each (profiles/step_timeline.py)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_step_tl}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 10 --warmup 2 --settle-ms 50 --no-side-legs --no-cpu --no-ic-leg > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
CSV=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/step_timeline.py $CSV > $OUT/${TAG}.txt
rm -rf $OUT/${TAG}_trace
