#!/bin/bash
# Iteration runs of one piece (parity tests first, then its timing script); results under gpurun_out/.
# Usage: bash scripts/gr.sh 'bash scripts/gpu_kernel.sh <k1|k2|k2sweep|k2calls|reserve|gzip|gunzip|coalesce|full|regions|pcie|shard|sq <regex>|final>'
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
T="timeout -k 10"
PYT="python -u -m pytest -x -q --timeout 120 --timeout-method thread"
case ${1:-} in
  k1)      # encoder: parity, per-series timing, in-kernel cycle and clock stamps (profiling build)
    $T 150 $PYT tests/test_rans_gpu.py tests/test_zz_no_alloc.py > gpurun_out/k1_tests.log 2>&1 || { tail -40 gpurun_out/k1_tests.log; exit 1; }
    $T 300 python -u scripts/series_time.py > gpurun_out/k1_series.log 2>&1 || { tail -20 gpurun_out/k1_series.log; exit 1; }
    grep -E "QS|small|all|1xQS" gpurun_out/k1_series.log
    CRAMGPU_PROF=1 $T 200 python -u scripts/prof_blocks.py > gpurun_out/k1_prof.log 2>&1 || { tail -20 gpurun_out/k1_prof.log; exit 1; }
    grep "cramgpu prof" gpurun_out/k1_prof.log | tail -6 ;;
  k2)      # decoder: parity, per-series timing (CRAMGPU_DEC_WAVES_PER_CU etc. from the environment)
    $T 150 $PYT tests/test_rans_gpu.py tests/test_foreign_streams.py > gpurun_out/k2_tests.log 2>&1 || { tail -40 gpurun_out/k2_tests.log; exit 1; }
    $T 300 python -u scripts/decode_time.py > gpurun_out/k2_series.log 2>&1 || { tail -20 gpurun_out/k2_series.log; exit 1; }
    grep -E "o1|o0|all |1xQS|round trip" gpurun_out/k2_series.log ;;
  gzip)    # gzip writer / gunzip: parity, the timing script (BGZF blocks and series-like streams)
    $T 300 $PYT tests/test_gzip_gpu.py tests/test_gunzip_gpu.py > gpurun_out/t_gz.log 2>&1 || { tail -30 gpurun_out/t_gz.log; exit 1; }
    $T 300 python3 -u scripts/gzip_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gzip_wave.txt || { tail gpurun_out/gzip_wave.txt; exit 1; }
    cat gpurun_out/gzip_wave.txt ;;
  coalesce) # coalescing queue: parity, one container per submission at 4 and 8 hardware queues, kernel trace of the 64-thread case
    $T 200 $PYT tests/test_coalesce_gpu.py > gpurun_out/t_co.log 2>&1 || { tail -30 gpurun_out/t_co.log; exit 1; }
    for hq in 4 8; do
      GPU_MAX_HW_QUEUES=$hq CO_THREADS=${CO_THREADS:-1,16,64,128,256} CO_TICKETS=${CO_TICKETS:-64} $T 300 python3 -u scripts/coalesce_time.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/co_native_q$hq.txt || { tail -5 gpurun_out/co_native_q$hq.txt; exit 1; }
      cat gpurun_out/co_native_q$hq.txt
    done
    CO_THREADS=64 CO_TICKETS= bash scripts/gpu_co_trace.sh ;;
  pcie)    # the job-slot entry points at several job sizes
    $T 300 python3 -u scripts/pcie_time.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/pcie_1000.txt || { tail -20 gpurun_out/pcie_1000.txt; exit 1; }
    cat gpurun_out/pcie_1000.txt ;;
  full)    # kernel trace of the encode_full region of the bench
    bash scripts/gpu_full_trace.sh ;;
  shard)   # the full per-GPU shard of configs[3]
    $T 900 python3 -u scripts/c4_full_shard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_full_shard.txt || { tail -20 gpurun_out/c4_full_shard.txt; exit 1; }
    cat gpurun_out/c4_full_shard.txt ;;
  k2sweep) # decoder: parity once, then the per-series timing for (waves per CU : roles) settings, K2_SWEEP="5:1 5:0 ..."
    $T 150 $PYT tests/test_rans_gpu.py tests/test_foreign_streams.py > gpurun_out/k2_tests.log 2>&1 || { tail -40 gpurun_out/k2_tests.log; exit 1; }
    for cfg in ${K2_SWEEP:-5:1 5:0 6:1 8:1 4:1}; do
      w=${cfg%%:*}; r=${cfg##*:}
      echo "== waves per CU $w, roles $r"
      CRAMGPU_DEC_WAVES_PER_CU=$w CRAMGPU_DEC_ROLES=$r $T 200 python -u scripts/decode_time.py > gpurun_out/k2_series_${w}_${r}.log 2>&1 || { tail -20 gpurun_out/k2_series_${w}_${r}.log; exit 1; }
      grep -E "QS/o1|TS/o1|all |1xQS|round trip" gpurun_out/k2_series_${w}_${r}.log
    done ;;
  k2calls) # every rans_decode_kernel launch of the bench in time order: duration, what started just before it, what ran beside it
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof_k2
    $T 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k2 -o bench -- python3 -u $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof_k2.log 2>&1 || { tail -30 $R/gpurun_out/prof_k2.log; exit 1; }
    python3 - /tmp/prof_k2 <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
for i, r in enumerate(rows):
    if "rans_decode_kernel" in r[0]:
        beside = sorted({q[0][:40] for q in rows if q[1] < r[2] and q[2] > r[1] and q is not r})
        prev = rows[i - 1]
        print("K2 %.2f ms grid %d, gap before %.3f ms (after %s); beside: %s" % ((r[2] - r[1]) / 1e6, r[3], (r[1] - prev[2]) / 1e6, prev[0][:40], beside))
for r in db.execute("select name, count(*), avg(duration) from kernels where name like '%series_%' group by name"):
    print("%-70s calls %3d avg %.3f ms" % (r[0][:70], r[1], r[2] / 1e6))
PY
    ;;
  reserve) # decode leg of the bench for several LDS reserves beside the decoder (room for the gunzip of the GZIP series), RESERVES="24 40 56" KiB
    for r in ${RESERVES:-24 40 56}; do
      CRAMGPU_DEC_LDS_RESERVE=$((r * 1024)) $T 300 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-full > gpurun_out/dec_reserve_$r.json 2> gpurun_out/dec_reserve_$r.err || { tail -20 gpurun_out/dec_reserve_$r.err; exit 1; }
      python3 -c "import json,sys; d=json.load(open(sys.argv[1]))['decode']; print('reserve %s KiB: decode leg %.2f ms, K2 %.2f ms, gunzip done at %.2f ms' % (sys.argv[2], d['ms_per_step'], d['rans_decode_kernel_ms'], d['gunzip']['ms_from_leg_start']))" gpurun_out/dec_reserve_$r.json $r
    done ;;
  gunzip)  # gunzip kernels: parity (zlib is the oracle, both kernels), then scripts/gunzip_time.py (20 000 BGZF blocks; CRAMGPU_GUNZIP_LANES picks the kernel)
    $T 300 $PYT tests/test_gunzip_gpu.py tests/test_gzip_gpu.py > gpurun_out/t_gunzip.log 2>&1 || { tail -30 gpurun_out/t_gunzip.log; exit 1; }
    tail -1 gpurun_out/t_gunzip.log
    $T 300 python3 -u scripts/gunzip_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gunzip_time.txt || { tail gpurun_out/gunzip_time.txt; exit 1; }
    cat gpurun_out/gunzip_time.txt ;;
  regions) # the configs[0] round trip, then the bench's encode_full / decode_full / ingest_full without the CPU baseline and PCIe legs
    $T 300 $PYT tests/test_configs_gpu.py > gpurun_out/t_configs.log 2>&1 || { tail -40 gpurun_out/t_configs.log; exit 1; }
    tail -1 gpurun_out/t_configs.log
    $T 400 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > gpurun_out/full_line.json 2> gpurun_out/full_err.log || { tail -20 gpurun_out/full_err.log; exit 1; }
    python3 scripts/bench_summary.py gpurun_out/full_line.json ;;
  sq)      # SQ counter pass of the bench for one kernel regex: gpu_kernel.sh sq <regex>
    RE=${2:-gunzip}
    cd /tmp && export TMPDIR=/tmp
    rm -rf $R/gpurun_out/pmc_k
    timeout -s KILL 420 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-include-regex "$RE" -d $R/gpurun_out/pmc_k -o pmc --output-format csv -- python3 -u $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc_k.log 2>&1 || { echo "pass failed"; tail -20 $R/gpurun_out/pmc_k.log; exit 1; }
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
    ;;
  final)   # what the driver runs at round end: smoke, the default bench line; then configs[4] as one pipeline
    $T 200 python3 -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
    $T 600 python3 -u bench.py > gpurun_out/final_line.json 2> gpurun_out/final_err.log || { tail -20 gpurun_out/final_err.log; exit 1; }
    python3 scripts/bench_summary.py gpurun_out/final_line.json
    $T 400 python3 -u bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/final_c4.json 2> gpurun_out/final_c4_err.log || { tail -20 gpurun_out/final_c4_err.log; exit 1; }
    cut -c1-300 gpurun_out/final_c4.json ;;
  *) echo "usage: gpu_kernel.sh <k1|k2|k2sweep|k2calls|reserve|gzip|gunzip|coalesce|full|regions|pcie|shard|sq <regex>|final>"; exit 2 ;;
esac
