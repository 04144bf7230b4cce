#!/usr/bin/env python
"""Turns gpurun_out/prof/bench_results.db (rocprofv3 --kernel-trace --stats) and the --pmc CSVs into the small text
summaries kept under profiles/ (per round)."""
import collections, csv, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")  # on the GPU box: gpurun_out/summary
os.makedirs(out, exist_ok=True)
db = sqlite3.connect(os.path.join(ROOT, "gpurun_out/prof/bench_results.db"))
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), "
                        "max(workgroup_x), max(vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie (durations in ns; "
            "includes the untimed workload set-up kernels of torch)\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,Grid,Workgroup,ArchVGPR,LDS\n")
    for r in rows[:40]:
        f.write('"%s",%d,%d,%.1f,%d,%d,%.3f,%d,%d,%d,%d\n' % (r[0][:110], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(ROOT, "gpurun_out/pmc_%s/pmc_counter_collection.csv" % c)
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out, "%s_pmc_%s.csv" % (tag, c)), "w") as f:
        f.write("# rocprofv3 --pmc %s --kernel-include-regex (scripts/gpu_round.sh KERNELS) -- python3 bench.py --steps 1 "
                "--warmup 1 --no-cpu-baseline --no-pcie ; counter unit KB per dispatch (raw, uncorrected)\n" % c)
        f.write("Kernel,Dispatches,MeanKB,MinKB,MaxKB\n")
        for k, v in agg.items():
            f.write('"%s",%d,%.0f,%.0f,%.0f\n' % (k[:110], len(v), sum(v) / len(v), min(v), max(v)))
print(open(os.path.join(out, "%s_kernel_stats.csv" % tag)).read()[:1500])
