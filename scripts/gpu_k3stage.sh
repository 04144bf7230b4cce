#!/bin/bash
# K3 count pass with and without the LDS staging: parity tests, kernel durations, FETCH_SIZE
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_records_gpu.py tests/test_bench_gpu.py tests/test_configs_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/k3_tests.txt 2>&1 || { tail -20 gpurun_out/k3_tests.txt; exit 1; }
tail -2 gpurun_out/k3_tests.txt
cd /tmp && export TMPDIR=/tmp
for ST in 1 0; do
  export CRAMGPU_K3_STAGE=$ST
  rm -rf $R/gpurun_out/prof_k3s$ST
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k3s$ST -o b -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-decode > $R/gpurun_out/k3s$ST.log 2>&1 || { tail -20 $R/gpurun_out/k3s$ST.log; exit 1; }
  python3 - $R/gpurun_out/prof_k3s$ST/b_results.db $ST <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%extract_lane%' group by name"):
    print("stage=%s %-60s calls %d avg %.1f us min %.1f" % (sys.argv[2], r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3))
PY
  rm -rf $R/gpurun_out/pmc_k3s$ST
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "extract_lane" -d $R/gpurun_out/pmc_k3s$ST -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --no-decode > $R/gpurun_out/pmc_k3s$ST.log 2>&1 || { tail -20 $R/gpurun_out/pmc_k3s$ST.log; exit 1; }
  python3 - $R/gpurun_out/pmc_k3s$ST $ST <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("stage=%s FETCH_SIZE %-50s mean %.0f KB" % (sys.argv[2], k, sum(v) / len(v)))
PY
done
