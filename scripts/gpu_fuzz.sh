#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python3 -u scripts/fuzz_records_gpu.py 10 > gpurun_out/fuzz_records.log 2>&1 || { echo "records fuzz failed"; tail -30 gpurun_out/fuzz_records.log; exit 1; }
tail -3 gpurun_out/fuzz_records.log
