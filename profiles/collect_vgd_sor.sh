set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
python3 $R/profiles/vgd_sor_probe.py > $OUT/r05v_probe.json 2> $OUT/r05v_probe.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05v_trace -- python3 $R/profiles/vgd_sor_probe.py > /dev/null 2> $OUT/r05v_trace.err
find $OUT/r05v_trace -name "*kernel_stats.csv" -exec cp {} $OUT/r05v_kernel_stats.csv \;
rm -rf $OUT/r05v_trace
