#!/bin/bash
# new-kernel bring-up: smallest test first under a short limit, nothing runs after a failure
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 90 python -u -m pytest tests/test_rans_gpu.py -x -q --timeout 60 --timeout-method thread -k "kat1_hex" > gpurun_out/t1.log 2>&1 || { echo "t1 failed"; tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t4.log 2>&1 || { echo "t4 failed"; tail -40 gpurun_out/t4.log; exit 1; }
tail -2 gpurun_out/t4.log
timeout -k 10 300 python3 -u scripts/series_time.py > gpurun_out/series_time.log 2>&1 || { echo "series failed"; tail -30 gpurun_out/series_time.log; exit 1; }
cat gpurun_out/series_time.log
