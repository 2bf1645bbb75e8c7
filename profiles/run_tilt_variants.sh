#!/bin/bash
# A/B of library builds (PCP_HIP_LIBRARY) on the whole-run hull pass: 4 keyframes in flight (wall) and one (kernel totals).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  export PCP_HIP_LIBRARY=$R/pointcloudprocessor_amd/lib/$v
  echo "== $v"
  PCP_HPR_LANES=4 python3 $R/profiles/hpr_pass_probe.py 2>/dev/null
  PCP_HPR_LANES=4 python3 $R/profiles/hpr_pass_probe.py 2>/dev/null
  bash $R/profiles/collect_hpr_pass.sh var_$v > /dev/null 2>&1
  sed -n 2,6p $R/gpurun_out/var_${v}_kernels.txt
done
