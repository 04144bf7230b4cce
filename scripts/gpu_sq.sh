#!/bin/bash
# SQ counters of the rANS kernels in the bench command: one rocprofv3 --pmc pass (8 SQ counters fit one pass),
# program directly after "--".  Usage: gpu_sq.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sq}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$TAG
timeout -s KILL 420 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-include-regex "rans_encode|rans_decode" -d $R/gpurun_out/pmc_$TAG -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_$TAG.log 2>&1 || { echo "pmc $TAG failed"; tail -20 $R/gpurun_out/pmc_$TAG.log; exit 1; }
python3 - $R/gpurun_out/pmc_$TAG <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("  %-22s dispatches %d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
