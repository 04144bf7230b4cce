#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 400 python3 -u scripts/c5_time.py > gpurun_out/c5.log 2>&1 || { echo "c5 failed"; tail -30 gpurun_out/c5.log; exit 1; }
cat gpurun_out/c5.log
