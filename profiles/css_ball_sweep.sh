#!/bin/bash
# the streamed chain on the whole C3 map under different sizes of the trailing outlier removal's ball (PCP_SOR_BALL x (k + 1) rows
# by the voxel structure's density bound): second call of profiles/css_probe.py each.  WAVE_STATS=1: the flagged rows' tally
# (its atomics slow the run down several-fold: counts only)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_css_ball}
: > $OUT/${TAG}.log
for B in ${BALLS:-1.2 1.7 1.9 2.1 2.3 2.7}; do
  echo "== ball $B" >> $OUT/${TAG}.log
  PCP_SOR_BALL=$B PCP_SOR_WAVE_STATS=${WAVE_STATS:-0} python3 $R/profiles/css_probe.py 28 2 >> $OUT/${TAG}.log 2> $OUT/${TAG}_$B.err
  grep "n=2[0-9]*" $OUT/${TAG}_$B.err | head -2 | cut -c1-200 >> $OUT/${TAG}.log || true
  rm -f $OUT/${TAG}_$B.err
done
