#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py tests/test_bench_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -2 gpurun_out/t_one.log
bash scripts/gpu_k3.sh
