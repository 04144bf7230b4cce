#!/bin/bash
# the full regions: the configs[0] round-trip test, then the bench's encode_full / decode_full / ingest_full
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_configs_gpu.py > gpurun_out/t_configs.log 2>&1 || { tail -40 gpurun_out/t_configs.log; exit 1; }
tail -1 gpurun_out/t_configs.log
bash scripts/gpu_ingest.sh
