#!/bin/bash
# decoder bring-up: rANS GPU tests first, then decode timing; nothing runs after a failure
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 200 python -u -m pytest tests/test_rans_gpu.py tests/test_foreign_streams.py -m gpu -x -q --timeout 90 --timeout-method thread > gpurun_out/t_dec.log 2>&1 || { echo "rans tests failed"; tail -40 gpurun_out/t_dec.log; exit 1; }
tail -2 gpurun_out/t_dec.log
timeout -k 10 300 python3 -u scripts/decode_time.py > gpurun_out/decode_time.log 2>&1 || { echo "decode_time failed"; tail -30 gpurun_out/decode_time.log; exit 1; }
cat gpurun_out/decode_time.log
