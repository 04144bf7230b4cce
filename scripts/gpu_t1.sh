#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py tests/test_abi.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -3 gpurun_out/t_one.log
