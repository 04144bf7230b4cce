#!/bin/bash
# timeline of the asynchronous boundary: kernel + memory-copy trace of four 256-container jobs
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pcie_trace
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/pcie_trace -o t -- python3 -u $R/scripts/pcie_time.py 1000 256 > $R/gpurun_out/pcie_trace.log 2>&1 || { echo "trace failed"; tail -20 $R/gpurun_out/pcie_trace.log; exit 1; }
tail -2 $R/gpurun_out/pcie_trace.log
python3 - $R/gpurun_out/pcie_trace <<'PY'
import csv, sys
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(d + "/t_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "rans_encode" in n or "gather_copy" in n:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + ("rans_encode" if "rans_encode" in n else "gather_copy"), r.get("Stream_Id", "")))
for r in csv.DictReader(open(d + "/t_memory_copy_trace.csv")):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 200000:
        ev.append((s, e, "copy " + r["Direction"][12:], r["Stream_Id"]))
ev.sort()
enc = [e for e in ev if "rans_encode" in e[2]]
lo = enc[-4][0] - 30_000_000  # the last (timed) run: four jobs
run = [e for e in ev if e[0] >= lo]
t0 = run[0][0]
print("# rocprofv3 --kernel-trace --memory-copy-trace -- python3 scripts/pcie_time.py 1000 256 : the timed run (4 jobs of 256 / 232 containers,")
print("# cramgpu_rans_submit_packed, 4 outstanding); copies shorter than 0.2 ms (descriptors, lengths) left out")
print("%-28s %-6s %9s %9s %8s" % ("event", "stream", "start ms", "end ms", "dur ms"))
for s, e, n, st in run:
    print("%-28s %-6s %9.2f %9.2f %8.2f" % (n, st, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
