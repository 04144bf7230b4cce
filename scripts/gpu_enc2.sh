#!/bin/bash
# rANS encoder check: GPU parity tests, per-series timing, bench line
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_rans_gpu.py tests/test_bench_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_one.log 2>&1 || { echo "test failed"; tail -40 gpurun_out/t_one.log; exit 1; }
tail -2 gpurun_out/t_one.log
timeout -k 10 300 python -u scripts/series_time.py > gpurun_out/series.log 2>&1 || { echo "series failed"; tail -20 gpurun_out/series.log; exit 1; }
cat gpurun_out/series.log
timeout -k 10 400 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1 || { echo "bench failed"; tail -30 gpurun_out/bench_small.log; exit 1; }
tail -1 gpurun_out/bench_small.log
