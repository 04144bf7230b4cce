#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -2 gpurun_out/t_one.log
timeout -k 10 600 python -u scripts/pcie_time.py > gpurun_out/pcie.log 2>&1 || { echo "pcie failed"; tail -20 gpurun_out/pcie.log; exit 1; }
grep containers gpurun_out/pcie.log
