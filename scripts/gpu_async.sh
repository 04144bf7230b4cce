#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { echo "gpu tests failed"; tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
bash scripts/gpu_pcie.sh || exit 1
bash scripts/gpu_bench.sh
