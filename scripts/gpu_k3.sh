#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 200 python3 -u scripts/k3_time.py > gpurun_out/k3.log 2>&1 || { echo "k3_time failed"; tail -30 gpurun_out/k3.log; exit 1; }
tail -9 gpurun_out/k3.log
