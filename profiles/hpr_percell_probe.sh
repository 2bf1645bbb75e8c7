#!/bin/bash
# the whole-run hull pass of C3 by candidates per fine cell of the gnomonic grid (PCP_HPR_PER_CELL): wall with four keyframes in
# flight, kernel totals with one.  (Needs a build in which hpr_finish reads PCP_HPR_PER_CELL in place of kHprTargetPerCell: the
# shipped library has the constant.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in ${PER_CELL:-4 6 8 12 16}; do
  export PCP_HPR_PER_CELL=$n
  echo "== candidates per cell $n"
  PCP_HPR_LANES=4 python3 $R/profiles/hpr_pass_probe.py 2>/dev/null
  PCP_HPR_LANES=4 python3 $R/profiles/hpr_pass_probe.py 2>/dev/null
  bash $R/profiles/collect_hpr_pass.sh percell_$n > /dev/null 2>&1
  sed -n 2,12p $R/gpurun_out/percell_${n}_kernels.txt | cut -c1-100
done
