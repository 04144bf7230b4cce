#!/bin/bash
# round 2, first GPU call: micro-benchmark of the rotating chain, then the GPU test suite
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 120 ./scripts/ubench/rot_bench > gpurun_out/rot_bench.txt 2>&1 || { echo "rot_bench failed"; tail -5 gpurun_out/rot_bench.txt; exit 1; }
cat gpurun_out/rot_bench.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { echo "gpu tests failed"; tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -5 gpurun_out/t_gpu.log
