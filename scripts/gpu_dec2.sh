#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
for W in 4 5 6; do
  echo "== CRAMGPU_DEC_WAVES_PER_CU=$W (LDS share)"
  CRAMGPU_DEC_WAVES_PER_CU=$W timeout -k 10 200 python3 -u scripts/decode_time.py > gpurun_out/decode_time_w$W.log 2>&1 || { echo "decode_time failed"; tail -30 gpurun_out/decode_time_w$W.log; exit 1; }
  grep "QS\|TS\|all\|round" gpurun_out/decode_time_w$W.log
done
