#!/bin/bash
# A/B timing of experiment builds of the library: scripts/gpu_exp.sh name1 name2 ... (cramtools_amd/libcramgpu_<name>.so)
mkdir -p gpurun_out
set -o pipefail
for name in "$@"; do
  lib=cramtools_amd/libcramgpu_$name.so
  [ "$name" = base ] && lib=cramtools_amd/libcramgpu.so
  CRAMGPU_LIB=$PWD/$lib timeout -k 10 300 python -u scripts/series_time.py > gpurun_out/exp_$name.log 2>&1 || { echo "$name failed"; tail -20 gpurun_out/exp_$name.log; exit 1; }
  echo "== $name"; grep -E "QS|small|all|1xQS" gpurun_out/exp_$name.log
done
