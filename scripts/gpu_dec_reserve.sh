#!/bin/bash
# decode leg of the bench for several LDS reserves beside the decoder (room for the gunzip of the GZIP series)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for r in ${RESERVES:-24 40 56}; do
  CRAMGPU_DEC_LDS_RESERVE=$((r * 1024)) timeout -k 10 300 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-full > gpurun_out/dec_reserve_$r.json 2> gpurun_out/dec_reserve_$r.err || { tail -20 gpurun_out/dec_reserve_$r.err; exit 1; }
  python3 - gpurun_out/dec_reserve_$r.json $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["decode"]
print("reserve %s KiB: decode leg %.2f ms, K2 %.2f ms, gunzip done at %.2f ms" % (sys.argv[2], d["ms_per_step"], d["rans_decode_kernel_ms"], d["gunzip"]["ms_from_leg_start"]))
PY
done
