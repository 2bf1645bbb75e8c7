#!/bin/bash
# kernel stats of the streamed chain on the whole C3 map (profiles/css_probe.py)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_css}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/profiles/css_probe.py > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
rm -rf $OUT/${TAG}_trace
