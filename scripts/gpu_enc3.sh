#!/bin/bash
# rANS encoder iteration: parity tests, in-kernel cycle counters (prof build), per-series timing
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -2 gpurun_out/t_one.log
timeout -k 10 200 python -u scripts/prof_blocks.py > gpurun_out/prof_blocks.log 2>&1 || { echo "prof failed"; tail -20 gpurun_out/prof_blocks.log; exit 1; }
grep -v amdgpu.ids gpurun_out/prof_blocks.log | tail -4
timeout -k 10 300 python -u scripts/series_time.py > gpurun_out/series.log 2>&1 || { echo "series failed"; tail -20 gpurun_out/series.log; exit 1; }
grep -E "QS|small|all|1xQS" gpurun_out/series.log
