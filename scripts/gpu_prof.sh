#!/bin/bash
# bench line + rocprofv3 kernel trace of the same command (summaries are copied to profiles/ afterwards)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -u $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/bench_full.log 2>&1 || { echo "bench failed"; tail -30 $R/gpurun_out/bench_full.log; exit 1; }
tail -1 $R/gpurun_out/bench_full.log
rm -rf $R/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof.log 2>&1 || { echo "rocprof failed"; tail -30 $R/gpurun_out/prof.log; exit 1; }
tail -1 $R/gpurun_out/prof.log
find $R/gpurun_out/prof -name "*stats*" | head
