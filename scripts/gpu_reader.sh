#!/bin/bash
# data-series reader: parity, then the bench's decode leg
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_series_gpu.py tests/test_configs_gpu.py > gpurun_out/t_reader.log 2>&1 || { tail -40 gpurun_out/t_reader.log; exit 1; }
tail -1 gpurun_out/t_reader.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_r
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-full > $R/gpurun_out/prof_r.log 2>&1 || { tail -30 $R/gpurun_out/prof_r.log; exit 1; }
python3 - $R/gpurun_out/prof_r <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
for r in db.execute("select name, count(*), avg(duration) from kernels where name like '%series_%' or name like '%rans_%' group by name"):
    print("%-70s calls %3d avg %.3f ms" % (r[0][:70], r[1], r[2] / 1e6))
PY
tail -c 600 $R/gpurun_out/prof_r.log | head -c 300
