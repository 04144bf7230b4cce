#!/bin/bash
# staged first contact with the GPU: each step under its own time limit, nothing runs after a failure
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 200 python -u -m pytest tests/test_rans_gpu.py -x -v --timeout 120 --timeout-method thread -k "kat1_hex or empty_short or large_quality" > gpurun_out/t1.log 2>&1 || { echo "t1 failed"; tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py -x -v --timeout 120 --timeout-method thread -k "not kat1_hex and not empty_short and not large_quality" > gpurun_out/t2.log 2>&1 || { echo "t2 failed"; tail -30 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python -u -m pytest tests/test_records_gpu.py -x -v --timeout 120 --timeout-method thread > gpurun_out/t3.log 2>&1 || { echo "t3 failed"; tail -30 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
