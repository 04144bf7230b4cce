#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
for W in 2 4 8 16 32; do
  echo "== CRAMGPU_LANE_WGS_PER_CU=$W"
  CRAMGPU_LANE_WGS_PER_CU=$W timeout -k 10 200 python3 -u scripts/k3_time.py > gpurun_out/k3_w$W.log 2>&1 || { echo "k3_time failed"; tail -30 gpurun_out/k3_w$W.log; exit 1; }
  tail -4 gpurun_out/k3_w$W.log
done
