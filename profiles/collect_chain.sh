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
# A fourth pass with TA_* / TCP_* counters was REJECTED, not hung (gpurun_out/chain1_p4.err of round 2):
# rocprofiler_create_counter_config failed with error 38, "Request exceeds the capabilities of the hardware to collect" --
# the set asked for more counters than a TA / TCP block has slots for in one pass -- and rocprofv3 then aborted at the
# first dispatch.  Nothing in the library was at fault.  Such counters need passes of their own that fit the block
# (rocprofv3 --list-avail shows the per-block capacity); none is collected here, the SQ counters answer what was asked.
cd $R
python3 profiles/summarise_chain.py $OUT/${TAG}_pmc.json $OUT/${TAG}_trace $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3
python3 profiles/build_stamp.py $OUT/${TAG}_pmc.json > /dev/null  # the library these counters belong to (bench.py checks it)
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3 $OUT/${TAG}_p4 $OUT/${TAG}_p5 $OUT/${TAG}_trace
