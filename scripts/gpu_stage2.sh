#!/bin/bash
# records tests, whole GPU suite, then a small and the full bench; nothing runs after a failure
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_records_gpu.py -x -v --timeout 120 --timeout-method thread > gpurun_out/t3.log 2>&1 || { echo "t3 failed"; tail -40 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
timeout -k 10 600 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/t4.log 2>&1 || { echo "t4 failed"; tail -40 gpurun_out/t4.log; exit 1; }
tail -3 gpurun_out/t4.log
timeout -k 10 300 python -u bench.py --reads 1000000 --ref-len 6400000 --steps 2 --warmup 1 > gpurun_out/bench_small.log 2>&1 || { echo "bench small failed"; tail -40 gpurun_out/bench_small.log; exit 1; }
tail -2 gpurun_out/bench_small.log
timeout -k 10 600 python -u bench.py > gpurun_out/bench_full.log 2>&1 || { echo "bench full failed"; tail -40 gpurun_out/bench_full.log; exit 1; }
tail -2 gpurun_out/bench_full.log
