#!/bin/bash
# lane-per-read record kernels against the number of resident workgroups per CU: durations and HBM traffic
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/k3_occ.txt
for W in 32 8 6 4 3; do
  export CRAMGPU_LANE_WGS_PER_CU=$W
  rm -rf /tmp/prof_k3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k3 -o b -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /tmp/k3.log 2>&1 || { tail -20 /tmp/k3.log; exit 1; }
  python3 - /tmp/prof_k3/b_results.db $W >> $R/gpurun_out/k3_occ.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%_lane_%' group by name"):
    print("wgs/cu=%-3s %-48s calls %d avg %8.1f us min %8.1f" % (sys.argv[2], r[0][21:69], r[1], r[2] / 1e3, r[3] / 1e3))
PY
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_k3
    timeout -s KILL 300 rocprofv3 --pmc $C --kernel-include-regex "_lane_" -d /tmp/pmc_k3 -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > /tmp/pmc_k3.log 2>&1 || { tail -20 /tmp/pmc_k3.log; exit 1; }
    python3 - /tmp/pmc_k3 $W $C >> $R/gpurun_out/k3_occ.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"][21:69]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("wgs/cu=%-3s %-10s %-48s mean %9.0f KB" % (sys.argv[2], sys.argv[3], k, sum(v) / len(v)))
PY
  done
done
cat $R/gpurun_out/k3_occ.txt
