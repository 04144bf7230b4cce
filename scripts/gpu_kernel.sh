#!/bin/bash
# Iteration runs of one piece (parity tests first, then its timing script); results under gpurun_out/.
# Usage: bash scripts/gr.sh 'bash scripts/gpu_kernel.sh <k1|k2|gzip|coalesce|full|pcie|shard>'
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
T="timeout -k 10"
PYT="python -u -m pytest -x -q --timeout 120 --timeout-method thread"
case ${1:-} in
  k1)      # encoder: parity, per-series timing, in-kernel cycle and clock stamps (profiling build)
    $T 150 $PYT tests/test_rans_gpu.py tests/test_zz_no_alloc.py > gpurun_out/k1_tests.log 2>&1 || { tail -40 gpurun_out/k1_tests.log; exit 1; }
    $T 300 python -u scripts/series_time.py > gpurun_out/k1_series.log 2>&1 || { tail -20 gpurun_out/k1_series.log; exit 1; }
    grep -E "QS|small|all|1xQS" gpurun_out/k1_series.log
    CRAMGPU_PROF=1 $T 200 python -u scripts/prof_blocks.py > gpurun_out/k1_prof.log 2>&1 || { tail -20 gpurun_out/k1_prof.log; exit 1; }
    grep "cramgpu prof" gpurun_out/k1_prof.log | tail -6 ;;
  k2)      # decoder: parity, per-series timing (CRAMGPU_DEC_WAVES_PER_CU etc. from the environment)
    $T 150 $PYT tests/test_rans_gpu.py tests/test_foreign_streams.py > gpurun_out/k2_tests.log 2>&1 || { tail -40 gpurun_out/k2_tests.log; exit 1; }
    $T 300 python -u scripts/decode_time.py > gpurun_out/k2_series.log 2>&1 || { tail -20 gpurun_out/k2_series.log; exit 1; }
    grep -E "o1|o0|all |1xQS|round trip" gpurun_out/k2_series.log ;;
  gzip)    # gzip writer / gunzip: parity, the timing script (BGZF blocks and series-like streams)
    $T 300 $PYT tests/test_gzip_gpu.py tests/test_gunzip_gpu.py > gpurun_out/t_gz.log 2>&1 || { tail -30 gpurun_out/t_gz.log; exit 1; }
    $T 300 python3 -u scripts/gzip_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gzip_wave.txt || { tail gpurun_out/gzip_wave.txt; exit 1; }
    cat gpurun_out/gzip_wave.txt ;;
  coalesce) # coalescing queue: parity, one container per submission at 4 and 8 hardware queues, kernel trace of the 64-thread case
    $T 200 $PYT tests/test_coalesce_gpu.py > gpurun_out/t_co.log 2>&1 || { tail -30 gpurun_out/t_co.log; exit 1; }
    for hq in 4 8; do
      GPU_MAX_HW_QUEUES=$hq CO_THREADS=${CO_THREADS:-1,16,64,128,256} CO_TICKETS=${CO_TICKETS:-64} $T 300 python3 -u scripts/coalesce_time.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/co_native_q$hq.txt || { tail -5 gpurun_out/co_native_q$hq.txt; exit 1; }
      cat gpurun_out/co_native_q$hq.txt
    done
    CO_THREADS=64 CO_TICKETS= bash scripts/gpu_co_trace.sh ;;
  pcie)    # the job-slot entry points at several job sizes
    $T 300 python3 -u scripts/pcie_time.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/pcie_1000.txt || { tail -20 gpurun_out/pcie_1000.txt; exit 1; }
    cat gpurun_out/pcie_1000.txt ;;
  full)    # kernel trace of the encode_full region of the bench
    bash scripts/gpu_full_trace.sh ;;
  shard)   # the full per-GPU shard of configs[3]
    $T 900 python3 -u scripts/c4_full_shard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_full_shard.txt || { tail -20 gpurun_out/c4_full_shard.txt; exit 1; }
    cat gpurun_out/c4_full_shard.txt ;;
  *) echo "usage: gpu_kernel.sh <k1|k2|gzip|coalesce|full|pcie|shard>"; exit 2 ;;
esac
