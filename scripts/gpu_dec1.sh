#!/bin/bash
# decoder round: parity tests of the rANS paths, then the decode timing on the bench corpus
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_rans_gpu.py tests/test_foreign_streams.py -x -q --timeout 300 --timeout-method thread > gpurun_out/dec_tests.txt 2>&1 || { tail -30 gpurun_out/dec_tests.txt; exit 1; }
tail -3 gpurun_out/dec_tests.txt
timeout -k 10 300 python -u scripts/decode_time.py > gpurun_out/decode_time.txt 2>&1 || { tail -30 gpurun_out/decode_time.txt; exit 1; }
cat gpurun_out/decode_time.txt
