#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of `bench.py` (N = 1 default workload, side legs off) and three PMC passes of the same
#   command (SQ counters; FETCH_SIZE; WRITE_SIZE -- they do not fit one pass), summarised by summarise_pmc.py.
# Outputs land under gpurun_out/<tag>_*; copy the summaries into profiles/ and commit them.
set -e -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --settle-ms 50 --no-side-legs --no-cpu"
cd /tmp && export TMPDIR=/tmp
# the trace runs without the cache-resident projection launches: the k_project_frame row of the stats is then the 40 M-point
# (HBM) launches alone and its average is the figure roofline.avg_launch_ms must agree with
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
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
# the raw per-dispatch csv files are large: keep only the summaries
rm -rf $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
find $OUT/${TAG}_trace -name "*kernel_trace.csv" -delete
