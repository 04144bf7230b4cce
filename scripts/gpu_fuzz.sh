#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 500 python3 -u scripts/fuzz_gpu.py 7 > gpurun_out/fuzz.log 2>&1 || { echo "fuzz failed"; tail -20 gpurun_out/fuzz.log; exit 1; }
tail -3 gpurun_out/fuzz.log
