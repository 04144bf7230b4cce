#!/usr/bin/env python
"""Times the rANS decode kernel on the bench corpus (encode -> gather -> decode), per series, and checks the round trip."""
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
    n_cont, ns, n = w["n_cont"], len(bench.SERIES), w["n_blocks"]
    st = bench.Step(dev, eng, w)
    st.run(False)
    torch.cuda.synchronize()
    lens = st.d_len.cpu().numpy().astype(np.int64)
    offs = st.d_pack_off.cpu().numpy().astype(np.int64)
    raw = np.array([w["blocks"][i].in_len for i in range(n)], np.int64)
    in_off = np.array([w["blocks"][i].in_off for i in range(n)], np.int64)
    d_raw = torch.zeros(w["d_in"].numel(), dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def run(idx, label):
        m = len(idx)
        blocks = (_lib.Block * m)()
        for i, b in enumerate(idx):
            blocks[i] = _lib.Block(int(offs[b]), int(in_off[b]), int(lens[b]), int(raw[b]), 0, 0)
        d_len = torch.zeros(m, dtype=torch.int32, device=dev)
        ms = []
        for rep in range(3):
            ev[0].record()
            eng.decode_dev(st.d_pack, d_raw, blocks, m, d_len)
            ev[1].record()
            torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1]))
        got = d_len.cpu().numpy()
        ok = bool((got == raw[idx]).all())
        print("%-6s blocks %6d raw %10.2f MB comp %9.2f MB  ms %s  lens_ok %s" % (label, m, raw[idx].sum() / 1e6, lens[idx].sum() / 1e6, ["%.3f" % x for x in ms], ok), flush=True)
    for k, (name, order) in enumerate(bench.SERIES):
        run(np.array([c * ns + k for c in range(n_cont)]), "%s/o%d" % (name, order))
    run(np.arange(n), "all")
    same = bool(torch.equal(d_raw[:w["d_in"].numel() - 16], w["d_in"][:-16]))
    print("round trip of all %d blocks identical to the raw arena: %s" % (n, same), flush=True)
    run(np.array([0]), "1xQS")
    eng.close()
    if not same:
        sys.exit(1)

if __name__ == "__main__":
    main()
