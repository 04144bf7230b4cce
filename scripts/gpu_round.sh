#!/bin/bash
# The round's evidence in one call (every step under its own time limit, nothing runs after a failure):
#   1. full GPU suite   2. bench line   3. rocprofv3 --kernel-trace --stats of the bench command
#   4. one --pmc pass per HBM counter (FETCH_SIZE, WRITE_SIZE do not fit one pass)   5. one SQ-counter pass of the rANS kernels
#   6. summary: the small per-kernel summaries (gpurun_out/summary/) — copy them into profiles/ here afterwards
# Usage: bash scripts/gr.sh 'bash scripts/gpu_round.sh [tests|bench|trace|pmc|sq ...]'   (default: all five)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
STEPS=${@:-tests bench trace pmc sq summary}
KERNELS="rans_encode|rans_decode|extract_|gather_copy|restore_|series_|gunzip|gzip|bam_|cigar_|mates_|mate_|content_crc|slice_md5"
for step in $STEPS; do
  case $step in
    tests)
      cd $R
      timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { echo "gpu tests failed"; tail -40 gpurun_out/t_all.log; exit 1; }
      tail -1 gpurun_out/t_all.log ;;
    bench)
      cd $R
      timeout -k 10 600 python3 -u bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log || { echo "bench failed"; tail -30 gpurun_out/bench_err.log; exit 1; }
      cut -c1-400 gpurun_out/bench_line.json ;;
    trace)
      cd /tmp && export TMPDIR=/tmp
      rm -rf $R/gpurun_out/prof
      timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof.log 2>&1 || { echo "rocprof failed"; tail -30 $R/gpurun_out/prof.log; exit 1; }
      find $R/gpurun_out/prof -name "*.db" | head -2 ;;
    pmc)
      cd /tmp && export TMPDIR=/tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        rm -rf $R/gpurun_out/pmc_$C
        timeout -s KILL 420 rocprofv3 --pmc $C --kernel-include-regex "$KERNELS" -d $R/gpurun_out/pmc_$C -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -20 $R/gpurun_out/pmc_$C.log; exit 1; }
        find $R/gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1
      done ;;
    sq)
      cd /tmp && export TMPDIR=/tmp
      rm -rf $R/gpurun_out/pmc_sq
      timeout -s KILL 420 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-include-regex "rans_encode|rans_decode|gzip_wave" -d $R/gpurun_out/pmc_sq -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_sq.log 2>&1 || { echo "sq pass failed"; tail -20 $R/gpurun_out/pmc_sq.log; exit 1; }
      python3 - $R/gpurun_out/pmc_sq > $R/gpurun_out/sq_counters.txt <<'PY'
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
      cat $R/gpurun_out/sq_counters.txt ;;
    summary)  # the small summaries of the trace and PMC passes (what profiles/ keeps); the trace database stays on the box
      cd $R
      python3 scripts/export_profile.py ${ROUND_TAG:-r03} gpurun_out/summary > /dev/null || { echo "summary failed"; exit 1; }
      rm -rf gpurun_out/prof
      ls gpurun_out/summary ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
