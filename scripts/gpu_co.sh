#!/bin/bash
# coalescing queue experiments: batch slots x hardware queues x hold
mkdir -p gpurun_out
set -o pipefail
export CO_THREADS=64 CO_TICKETS=64,128
for cfg in "3 8 300" "4 8 100" "6 8 300" "6 8 100" "6 16 300" "12 16 300"; do
  set -- $cfg
  echo "## batch slots $1, GPU_MAX_HW_QUEUES=$2, hold $3 us" >> gpurun_out/co_exp.txt
  CRAMGPU_COALESCE_HOLD_US=$3 GPU_MAX_HW_QUEUES=$2 timeout -k 10 120 python3 -u scripts/coalesce_time.py 256 $1 2>&1 | grep cramgpu_rans >> gpurun_out/co_exp.txt || { echo failed; tail -5 gpurun_out/co_exp.txt; exit 1; }
done
cat gpurun_out/co_exp.txt
