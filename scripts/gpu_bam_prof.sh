#!/bin/bash
# kernel trace of the bench with the assembly / BAM stage: durations of the bam.hip kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bam
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_bam -o b -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /tmp/bam.log 2>&1 || { tail -20 /tmp/bam.log; exit 1; }
python3 - /tmp/prof_bam/b_results.db > $R/gpurun_out/bam_kernels.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%bam_%' or name like '%cigar%' or name like '%mates%' or name like '%scan%' group by name order by avg(duration) desc"):
    print("%-90s calls %3d avg %9.1f us min %9.1f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3))
PY
cat $R/gpurun_out/bam_kernels.txt
