#!/bin/bash
# every rans_decode_kernel launch of the bench in time order (duration, what ran beside it)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k2
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k2 -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof_k2.log 2>&1 || { tail -30 $R/gpurun_out/prof_k2.log; exit 1; }
python3 - /tmp/prof_k2 <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
for i, r in enumerate(rows):
    if "rans_decode_kernel" in r[0]:
        beside = sorted({q[0][:40] for q in rows if q[1] < r[2] and q[2] > r[1] and q is not r})
        prev = rows[i - 1]
        print("K2 %.2f ms grid %d, gap before %.3f ms (after %s); beside: %s" % ((r[2] - r[1]) / 1e6, r[3], (r[1] - prev[2]) / 1e6, prev[0][:40], beside))
PY
