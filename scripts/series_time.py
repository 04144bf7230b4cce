#!/usr/bin/env python
"""Times the rANS encode kernel per data series of the bench workload (one launch per series) — a profiling aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cramtools_amd import _lib, external_compression as xc

def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = "cuda:0"
    eng = xc.Engine(0)
    w = bench.build_workload(dev, reads, reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads))
    n_cont, ns = w["n_cont"], len(bench.SERIES)
    d_out = torch.empty(w["out_bytes"], dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def run(idx, label):
        n = len(idx)
        blocks = (_lib.Block * n)()
        for i, b in enumerate(idx):
            blocks[i] = w["blocks"][b]
        d_len = torch.zeros(n, dtype=torch.int32, device=dev)
        d_pos = torch.zeros(n, dtype=torch.int64, device=dev)
        ms = []
        for rep in range(3):
            ev[0].record()
            eng.encode_dev(w["d_in"], d_out, blocks, n, d_len, d_pos)
            ev[1].record()
            torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1]))
        raw = sum(blocks[i].in_len for i in range(n))
        comp = int(d_len.clamp(min=0).sum().item())
        print("%-6s blocks %6d raw %10.2f MB comp %9.2f MB  ms %s  (avg block %d B)" % (label, n, raw / 1e6, comp / 1e6, ["%.3f" % m for m in ms], raw // max(n, 1)), flush=True)
    for k, (name, order) in enumerate(bench.SERIES):
        run([c * ns + k for c in range(n_cont)], "%s/o%d" % (name, order))
    run([c * ns + k for c in range(n_cont) for k in range(1, ns)], "small")
    run(list(range(n_cont * ns)), "all")
    run([0], "1xQS")
    eng.close()

if __name__ == "__main__":
    main()
