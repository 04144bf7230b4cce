#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_rans_gpu.py tests/test_foreign_streams.py tests/test_coalesce_gpu.py tests/test_series_gpu.py tests/test_zz_no_alloc.py > gpurun_out/t_k2.log 2>&1 || { tail -40 gpurun_out/t_k2.log; exit 1; }
tail -1 gpurun_out/t_k2.log
bash scripts/gpu_k2_calls.sh
