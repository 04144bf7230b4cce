#!/bin/bash
# gzip writer: tests, the timing script, and the bench's sizes / full-container regions
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gzip_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t_gz.log 2>&1 || { tail -30 gpurun_out/t_gz.log; exit 1; }
tail -1 gpurun_out/t_gz.log
timeout -k 10 300 python3 -u scripts/gzip_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gzip_wave.txt || { tail gpurun_out/gzip_wave.txt; exit 1; }
grep -E "sample|ms =" gpurun_out/gzip_wave.txt | awk 'NR%4==3 || /sample/'
timeout -k 10 400 python3 -u bench.py --no-cpu-baseline --no-pcie --steps 3 > gpurun_out/bench_full_line.json 2> gpurun_out/bench_full_err.log || { tail -5 gpurun_out/bench_full_err.log; exit 1; }
python3 -c "
import json
d = json.loads(open('gpurun_out/bench_full_line.json').read().strip().splitlines()[-1])
g = d['decode']['gzip_writer']; b = d['decode']['assembly']['bgzf']
print('step', round(d['ms_per_step'], 2), 'encode_full', round(d['encode_full']['ms'], 2), 'decode_full', round(d['decode_full']['ms'], 2))
print('series: %.2f ms %d B (zlib5 %d), cut8k %.2f ms %d B; bgzf %.1f ms %d B' % (g['ms'], g['gz_bytes'], g['zlib_level5_bytes'], g['ms_cut_8k'], g['gz_bytes_cut_8k'], b['ms'], b['bgzf_bytes']))
"
