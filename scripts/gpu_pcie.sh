#!/bin/bash
mkdir -p gpurun_out
set -o pipefail
for B in 4 16 64 256; do
  timeout -k 10 200 python3 -u scripts/pcie_time.py $B > gpurun_out/pcie_$B.log 2>&1 || { echo "pcie failed"; tail -20 gpurun_out/pcie_$B.log; exit 1; }
  tail -1 gpurun_out/pcie_$B.log
done
