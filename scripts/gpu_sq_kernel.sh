#!/bin/bash
# SQ counter pass of the bench for one kernel regex: bash scripts/gpu_sq_kernel.sh <regex> [counters...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
RE=${1:-gunzip}; shift
CTR=${@:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_k
timeout -s KILL 420 rocprofv3 --pmc $CTR --kernel-include-regex "$RE" -d $R/gpurun_out/pmc_k -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_k.log 2>&1 || { echo "pass failed"; tail -20 $R/gpurun_out/pmc_k.log; exit 1; }
python3 - $R/gpurun_out/pmc_k <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        agg[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("  %-22s dispatches %d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
