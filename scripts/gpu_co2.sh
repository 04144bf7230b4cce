#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
export CO_THREADS=16,64,128,256 CO_TICKETS=64
for hq in 4 8; do
  GPU_MAX_HW_QUEUES=$hq timeout -k 10 200 python3 -u scripts/coalesce_time.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/co_native_q$hq.txt || { echo failed; tail -5 gpurun_out/co_native_q$hq.txt; exit 1; }
  cat gpurun_out/co_native_q$hq.txt
done
