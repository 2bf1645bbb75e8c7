#!/bin/bash
# PMC passes over MLS alone (profiles/mls_probe.py): SQ counters, FETCH_SIZE, WRITE_SIZE (one block per pass), summarised per
# kernel by summarise_chain.py.  Run through gpurun from the repo root:  bash profiles/collect_mls.sh <tag>
set -e -o pipefail
TAG=${1:-r03_mls}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
CMD="python3 $R/profiles/mls_probe.py"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- $CMD > $OUT/${TAG}_probe.json 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/${TAG}_p1 -- $CMD > /dev/null 2> $OUT/${TAG}_p1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_p2 -- $CMD > /dev/null 2> $OUT/${TAG}_p2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_p3 -- $CMD > /dev/null 2> $OUT/${TAG}_p3.err
cd $R
python3 profiles/summarise_chain.py $OUT/${TAG}_pmc.json $OUT/${TAG}_trace $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3
python3 profiles/build_stamp.py $OUT/${TAG}_pmc.json > /dev/null  # the library these counters belong to (bench.py checks it)
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3 $OUT/${TAG}_trace
