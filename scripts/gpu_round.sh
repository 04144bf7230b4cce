#!/bin/bash
# decode timing, then bench + kernel trace, then PMC passes; nothing runs after a failure
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
timeout -k 10 300 python3 -u scripts/decode_time.py > gpurun_out/decode_time.log 2>&1 || { echo "decode_time failed"; tail -30 gpurun_out/decode_time.log; exit 1; }
cat gpurun_out/decode_time.log
bash scripts/gpu_prof.sh || exit 1
bash scripts/gpu_pmc.sh || exit 1
