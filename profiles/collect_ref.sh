#!/bin/bash
# PMC passes of the timed step with the reference's camera (4096x3000): what the colour pass fetches per candidate.
set -e -o pipefail
TAG=${1:-r02_ref}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BENCH="python3 $R/bench.py --camera ref --steps 5 --warmup 2 --no-side-legs --no-cpu --roofline-launches 8 --roofline-points 10000000"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d $OUT/${TAG}_pmc_sq -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_pmc_tcc -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_tcc.err || true
cd $R
python3 profiles/summarise_pmc.py $OUT/${TAG}_pmc.json "rocprofv3 --pmc {SQ_* | FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} -- python3 bench.py --camera ref --steps 5 --warmup 2 --no-side-legs --no-cpu (separate passes)" 10000000 256 10000000 $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_tcc
rm -rf $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_tcc
