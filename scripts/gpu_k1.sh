#!/bin/bash
# K1 iteration: encoder parity tests, per-series timing, in-kernel cycle counters (profiling build)
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 150 python -u -m pytest tests/test_rans_gpu.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/k1_tests.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/k1_tests.log; exit 1; }
tail -2 gpurun_out/k1_tests.log
timeout -k 10 300 python -u scripts/series_time.py > gpurun_out/k1_series.log 2>&1 || { echo "series failed"; tail -20 gpurun_out/k1_series.log; exit 1; }
grep -E "QS|small|all|1xQS" gpurun_out/k1_series.log
CRAMGPU_PROF=1 timeout -k 10 200 python -u scripts/prof_blocks.py > gpurun_out/k1_prof.log 2>&1 || { echo "prof failed"; tail -20 gpurun_out/k1_prof.log; exit 1; }
grep -v amdgpu.ids gpurun_out/k1_prof.log | tail -6
