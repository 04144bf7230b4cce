#!/bin/bash
# gunzip kernel: parity (zlib is the oracle), then scripts/gunzip_time.py (20 000 BGZF blocks of BAM-like records)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
T="timeout -k 10"
$T 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gunzip_gpu.py tests/test_gzip_gpu.py > gpurun_out/t_gunzip.log 2>&1 || { tail -30 gpurun_out/t_gunzip.log; exit 1; }
tail -1 gpurun_out/t_gunzip.log
$T 300 python3 -u scripts/gunzip_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gunzip_time.txt || { tail gpurun_out/gunzip_time.txt; exit 1; }
cat gpurun_out/gunzip_time.txt
