#!/bin/bash
# the encoder behind the writer (cramgpu_rans_encode_series_dev): parity, then the bench's step
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_series_gpu.py tests/test_rans_gpu.py tests/test_zz_no_alloc.py > gpurun_out/t_series.log 2>&1 || { tail -40 gpurun_out/t_series.log; exit 1; }
tail -1 gpurun_out/t_series.log
timeout -k 10 400 python3 -u bench.py --steps 5 --warmup 2 --no-pcie --no-full > gpurun_out/step_line.json 2> gpurun_out/step_err.log || { tail -20 gpurun_out/step_err.log; exit 1; }
python3 scripts/bench_summary.py gpurun_out/step_line.json
