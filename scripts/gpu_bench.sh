#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -u bench.py --steps 3 --warmup 1 > gpurun_out/bench_full.log 2>&1 || { echo "bench failed"; tail -30 gpurun_out/bench_full.log; exit 1; }
tail -1 gpurun_out/bench_full.log
