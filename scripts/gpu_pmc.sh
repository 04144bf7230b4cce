#!/bin/bash
# HBM traffic counters of the bench's kernels: one rocprofv3 --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not
# fit one pass), program directly after "--".
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout -s KILL 420 rocprofv3 --pmc $C --kernel-include-regex "rans_encode|rans_decode|extract_|gather_copy|restore_|series_|gunzip|bam_|cigar_|mates_" -d $R/gpurun_out/pmc_$C -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -20 $R/gpurun_out/pmc_$C.log; exit 1; }
  find $R/gpurun_out/pmc_$C -name "*.csv" | head -5
done
