#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_k3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k3 -o k3 --output-format csv -- python3 -u $R/scripts/k3_time.py > $R/gpurun_out/k3.log 2>&1 || { echo "failed"; tail -30 $R/gpurun_out/k3.log; exit 1; }
find $R/gpurun_out/prof_k3 -name "*kernel_stats*" | head -2
grep "extract\|scan_\|restore\|matrices\|assign" $(find $R/gpurun_out/prof_k3 -name "*kernel_stats.csv" | head -1) | cut -c1-200
