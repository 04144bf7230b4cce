#!/bin/bash
# decoder: parity once, then the per-series timing for several (waves per CU, roles) settings
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
T="timeout -k 10"
$T 150 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_rans_gpu.py tests/test_foreign_streams.py > gpurun_out/k2_tests.log 2>&1 || { tail -40 gpurun_out/k2_tests.log; exit 1; }
SWEEP=${K2_SWEEP:-5:0 6:1 8:1 4:1}
for cfg in $SWEEP; do
  w=${cfg%%:*}; r=${cfg##*:}
  echo "== waves per CU $w, roles $r"
  CRAMGPU_DEC_WAVES_PER_CU=$w CRAMGPU_DEC_ROLES=$r $T 200 python -u scripts/decode_time.py > gpurun_out/k2_series_${w}_${r}.log 2>&1 || { tail -20 gpurun_out/k2_series_${w}_${r}.log; exit 1; }
  grep -E "QS/o1|TS/o1|all |1xQS|round trip" gpurun_out/k2_series_${w}_${r}.log
done
