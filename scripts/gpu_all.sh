#!/bin/bash
# full GPU suite, then the bench line; nothing runs after a failure
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { echo "gpu tests failed"; tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
bash scripts/gpu_bench.sh
