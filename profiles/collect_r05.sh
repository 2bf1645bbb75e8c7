#!/bin/bash
# Every counter summary bench.py reads, collected on the library in the tree and stamped with its SHA-256
# (profiles/build_stamp.py): run through gpurun from the repo root, then copy gpurun_out/<tag>_* into profiles/ under the
# names bench.py expects (r05_pmc.json, r05_hpr_pmc.json, r05_mls_pmc.json, r05_chain_pmc.json) and commit them.
#   bash profiles/collect_r05.sh <tag>
set -e -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --settle-ms 50 --no-side-legs --no-cpu"
cd /tmp && export TMPDIR=/tmp
# (a) the bench command: kernel trace + stats (without the cache-resident projection launches: the k_project_frame row is then
# the 40 M-point launches alone) and four counter passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $BENCH --no-ic-leg > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
echo "trace done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_pmc_sq -- $BENCH > $OUT/${TAG}_pmc_sq_bench.json 2> $OUT/${TAG}_pmc_sq.err
echo "pmc sq done"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq2 -- $BENCH > $OUT/${TAG}_pmc_sq2_bench.json 2> $OUT/${TAG}_pmc_sq2.err
echo "pmc sq2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $BENCH > $OUT/${TAG}_pmc_fetch_bench.json 2> $OUT/${TAG}_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $BENCH > $OUT/${TAG}_pmc_write_bench.json 2> $OUT/${TAG}_pmc_write.err
echo "pmc write done"
cd $R
python3 profiles/summarise_pmc.py $OUT/${TAG}_pmc.json "rocprofv3 --pmc {SQ issue counters | SQ wait counters | FETCH_SIZE | WRITE_SIZE} --output-format csv -- python3 bench.py --steps 10 --warmup 2 --settle-ms 50 --no-side-legs --no-cpu (four separate passes)" 10000000 256 40000000 $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
python3 profiles/build_stamp.py $OUT/${TAG}_pmc.json > /dev/null
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
rm -rf $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_trace
echo "bench summaries done"
# (b) the enableMLS chain, (c) MLS alone, (d) the hull kernels (8 keyframes), (e) the whole hull pass with one keyframe in flight
bash profiles/collect_chain.sh ${TAG}_chain
echo "chain done"
bash profiles/collect_mls.sh ${TAG}_mls
echo "mls done"
bash profiles/collect_hpr_pmc.sh ${TAG}_hpr > $OUT/${TAG}_hpr_summary.txt
echo "hpr pmc done"
bash profiles/collect_hpr_pass.sh ${TAG}_hpr_pass
echo "hpr pass done"
