#!/bin/bash
# kernel trace of one bench run (no CPU baseline, no PCIe leg): per-kernel durations of the library's kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof.log 2>&1 || { echo "rocprof failed"; tail -30 $R/gpurun_out/prof.log; exit 1; }
tail -1 $R/gpurun_out/prof.log | cut -c1-300
python3 - $R/gpurun_out/prof/bench_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%anonymous%' group by name order by sum(duration) desc"))
for r in rows:
    print("%-70s calls %4d avg %10.1f us min %10.1f max %10.1f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
PY
