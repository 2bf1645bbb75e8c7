cd /tmp && export TMPDIR=/tmp
for w in 2 3 4; do
  export PCP_HIP_LIBRARY=$GRAFT_REPO_ROOT/pointcloudprocessor_amd/lib/libpcp_hip_wpe$w.so
  echo "== WPE $w"
  PCP_HPR_LANES=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04e_tr -- python3 $GRAFT_REPO_ROOT/profiles/hpr_pass_probe.py 2> $GRAFT_REPO_ROOT/gpurun_out/r04e_tr.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04e_tr -name "*kernel_trace.csv")
  python3 $GRAFT_REPO_ROOT/profiles/overlap.py $f 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r04e_ov$w.txt
  sed -n 1,5p $GRAFT_REPO_ROOT/gpurun_out/r04e_ov$w.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04e_tr
  PCP_HPR_LANES=4 python3 $GRAFT_REPO_ROOT/profiles/hpr_pass_probe.py 2>/dev/null
done
