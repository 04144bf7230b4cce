#!/bin/bash
# K5: parity tests of the record kernels, then the lane-per-read and the wave-per-read kernel on the configs[1] reads
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python -u -m pytest tests/test_records_gpu.py tests/test_configs_gpu.py tests/test_host_stripes_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/k5_tests.txt 2>&1 || { tail -30 gpurun_out/k5_tests.txt; exit 1; }
tail -3 gpurun_out/k5_tests.txt
{ echo "== lane per read (default)"; timeout -k 10 200 python -u scripts/k5_time.py; } > gpurun_out/k5_time.txt 2>&1 || { tail -20 gpurun_out/k5_time.txt; exit 1; }
{ echo "== wave per read (CRAMGPU_K5_LANE=0)"; CRAMGPU_K5_LANE=0 timeout -k 10 200 python -u scripts/k5_time.py; } >> gpurun_out/k5_time.txt 2>&1 || { tail -20 gpurun_out/k5_time.txt; exit 1; }
cat gpurun_out/k5_time.txt
