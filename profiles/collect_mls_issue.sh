#!/bin/bash
# What bounds k_mls_fit?  Gather form (PCP_MLS_TILE=0) and tile form of the same workload (profiles/mls_probe.py): the clock
# the kernel really runs at (GRBM_GUI_ACTIVE / duration), vector / LDS / scalar issue and waits.  One counter block per pass.
# Run through gpurun from the repo root:  bash profiles/collect_mls_issue.sh <tag>
set -e -o pipefail
TAG=${1:-r03l_mls_issue}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
CMD="python3 $R/profiles/mls_probe.py"
cd /tmp && export TMPDIR=/tmp
for MODE in 0 1; do
  export PCP_MLS_TILE=$MODE
  T=${TAG}_tile${MODE}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/${T}_trace -- $CMD > $OUT/${T}_probe.json 2> $OUT/${T}_trace.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${T}_p1 -- $CMD > /dev/null 2> $OUT/${T}_p1.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/${T}_p2 -- $CMD > /dev/null 2> $OUT/${T}_p2.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/${T}_p3 -- $CMD > /dev/null 2> $OUT/${T}_p3.err
  ( cd $R && python3 profiles/summarise_chain.py $OUT/${T}_pmc.json $OUT/${T}_trace $OUT/${T}_p1 $OUT/${T}_p2 $OUT/${T}_p3 )
  rm -rf $OUT/${T}_p1 $OUT/${T}_p2 $OUT/${T}_p3 $OUT/${T}_trace
done
