#!/bin/bash
# round-end rehearsal: GPU suite, smoke(), then bench + kernel trace + PMC passes
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { echo "gpu tests failed"; tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
bash scripts/gpu_profiles.sh
