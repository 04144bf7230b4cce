#!/bin/bash
# kernel trace of the encode_full / decode_full regions of the bench: start / end of every kernel of the LAST encode_full step
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/full_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/full_trace -o t -- python3 -u $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-bgzf ${FULL_ARGS:-} > $R/gpurun_out/full_trace.log 2>&1 || { echo "rocprof failed"; tail -30 $R/gpurun_out/full_trace.log; exit 1; }
f=$(find $R/gpurun_out/full_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/full_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last set_mate kernel marks the last encode_full step; print everything from the extract kernel before it to the gather after it
idx = [i for i, r in enumerate(rows) if "mate_key" in r["Kernel_Name"] or "set_mate" in r["Kernel_Name"] or "mateinfo" in r["Kernel_Name"].lower()]
names = sorted(set(r["Kernel_Name"][:60] for r in rows if "anonymous" in r["Kernel_Name"]))
print("# library kernels seen:", *names, sep="\n#   ")
if idx:
    last = idx[-1]
    start = last
    while start > 0 and "extract_lane_count" not in rows[start]["Kernel_Name"]:
        start -= 1
    t0 = int(rows[start]["Start_Timestamp"])
    for r in rows[start:]:
        if "anonymous" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]:
            continue
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if a > 60: break
        print("%-52s q%s %8.3f -> %8.3f ms (%.3f)" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:52], r["Queue_Id"], a, b, b - a))
PY
rm -rf $R/gpurun_out/full_trace
tail -70 $R/gpurun_out/full_trace.txt
