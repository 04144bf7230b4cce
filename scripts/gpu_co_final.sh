#!/bin/bash
# the coalescing queue over all 1000 containers of configs[1] (profiles/r03_pcie_inclusive.txt), then the bench line
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python3 -u scripts/coalesce_time.py 1000 > gpurun_out/coalesce_1000.txt 2>&1 || { echo "coalesce failed"; tail -20 gpurun_out/coalesce_1000.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/coalesce_1000.txt
timeout -k 10 200 python3 -u scripts/pcie_time.py 1000 > gpurun_out/pcie_1000.txt 2>&1 || { echo "pcie failed"; tail -20 gpurun_out/pcie_1000.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/pcie_1000.txt
timeout -k 10 400 python3 -u bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log || { echo "bench failed"; tail -20 gpurun_out/bench_err.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "K1", d["roofline"]["ms_per_launch"], "decode", d["decode"]["ms_per_step"], "K2", d["decode"]["rans_decode_kernel_ms"])
print(json.dumps(d["pcie_inclusive"])[:1500])
PY
