#!/bin/bash
# PMC passes over the enableMLS chain (profiles/chain_probe.py), one counter group per pass, summed per kernel by
# summarise_chain.py.  Run through gpurun from the repo root:  bash profiles/collect_chain.sh <tag>
set -e -o pipefail
TAG=${1:-chain}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
CMD="python3 $R/profiles/chain_probe.py"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- $CMD > /dev/null 2> $OUT/${TAG}_trace.err
echo "trace done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/${TAG}_p1 -- $CMD > /dev/null 2> $OUT/${TAG}_p1.err
echo "p1 done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/${TAG}_p2 -- $CMD > /dev/null 2> $OUT/${TAG}_p2.err
echo "p2 done"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_p3 -- $CMD > /dev/null 2> $OUT/${TAG}_p3.err
echo "p3 done"
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/${TAG}_p4 -- $CMD > /dev/null 2> $OUT/${TAG}_p4.err || echo "p4 failed"
echo "p4 done"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/${TAG}_p5 -- $CMD > /dev/null 2> $OUT/${TAG}_p5.err || echo "p5 failed"
echo "p5 done"
cd $R
python3 profiles/summarise_chain.py $OUT/${TAG}_pmc.json $OUT/${TAG}_trace $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3 $OUT/${TAG}_p4 $OUT/${TAG}_p5
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3 $OUT/${TAG}_p4 $OUT/${TAG}_p5 $OUT/${TAG}_trace
