#!/bin/bash
# K1 cycles against wall time: the profiling build (in-kernel cycle stamps) standing alone and inside the bench step
set -o pipefail
mkdir -p gpurun_out
export CRAMGPU_PROF=1
timeout -k 10 300 python -u scripts/prof_blocks.py > gpurun_out/k1_cycles_alone.txt 2>&1 || { tail gpurun_out/k1_cycles_alone.txt; exit 1; }
timeout -k 10 400 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-pcie > gpurun_out/k1_cycles_bench.txt 2>&1 || { tail gpurun_out/k1_cycles_bench.txt; exit 1; }
grep "cramgpu prof" gpurun_out/k1_cycles_alone.txt | cut -c1-110
grep "blocks " gpurun_out/k1_cycles_alone.txt
grep "cramgpu prof" gpurun_out/k1_cycles_bench.txt | cut -c1-110
tail -1 gpurun_out/k1_cycles_bench.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: ms_per_step', d['ms_per_step'], 'K1 ms_per_launch (mean of timed steps)', d['roofline']['ms_per_launch'])"
