#!/bin/bash
# the full regions of the bench (encode_full / decode_full / ingest_full) without the CPU baseline and PCIe legs
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 400 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/full_line.json 2> gpurun_out/full_err.log || { tail -20 gpurun_out/full_err.log; exit 1; }
python3 scripts/bench_summary.py gpurun_out/full_line.json
