#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_bench_gpu.py -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -5 gpurun_out/t_one.log
