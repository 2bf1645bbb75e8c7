#!/bin/bash
# The parity tests under the switches that select alternative forms of the same computation (every form must give the same
# results): run through gpurun from the repo root.   bash scripts/switch_matrix.sh > gpurun_out/switch_matrix.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # run <label> <env assignments...> -- <pytest args>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local out
  out=$(env "${envs[@]}" python3 -m pytest "$@" -x -q 2>&1 | tail -1)
  echo "$label [${envs[*]}]: $out"
}
run hull PCP_TILT_BUDGET=0 -- tests/test_hpr_gpu.py
run hull PCP_TILT_BUDGET=16 -- tests/test_hpr_gpu.py
run hull PCP_HPR_ONESTEP=0 -- tests/test_hpr_gpu.py
run hull PCP_HPR_ONESTEP=0 PCP_TILT_BUDGET=0 PCP_HPR_LANES=1 -- tests/test_hpr_gpu.py
run hull PCP_HPR_QUICK=0 PCP_HPR_RADIAL=0 -- tests/test_hpr_gpu.py
run hull PCP_HPR_TILT=0 -- tests/test_hpr_gpu.py
run hull PCP_HPR_LANES=8 PCP_TILT_BUDGET=8 -- tests/test_hpr_gpu.py tests/test_full_size_gpu.py
run colour PCP_RESULT_UNPERMUTE=2 -- tests/test_colour_gpu.py tests/test_golden_gpu.py tests/test_baseline_configs_gpu.py
run colour PCP_RESULT_UNPERMUTE=1 -- tests/test_colour_gpu.py tests/test_golden_gpu.py
run smooth PCP_SOR_CLUSTERED=1 -- tests/test_sor_gpu.py tests/test_mls_gpu.py tests/test_smooth_stream_gpu.py
run smooth PCP_SOR_CLUSTERED=0 -- tests/test_smooth_stream_gpu.py tests/test_mls_gpu.py
run smooth PCP_GRID_SPARSE=1 -- tests/test_smooth_stream_gpu.py tests/test_sor_gpu.py
run smooth PCP_CSS_HALO=1 -- tests/test_smooth_stream_gpu.py -k oracle
run smooth PCP_VGD_GRID=fit -- tests/test_smooth_stream_gpu.py
run hull PCP_HPR_DEBUG=1 -- tests/test_hpr_gpu.py
run smooth PCP_SOR_BALL=1.2 -- tests/test_smooth_stream_gpu.py
run smooth PCP_SOR_BALL=2.4 -- tests/test_smooth_stream_gpu.py tests/test_sor_gpu.py
