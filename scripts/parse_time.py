#!/usr/bin/env python
"""Times the reader's launches on the bench corpus with one series alive at a time (which stream costs what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cramtools_amd import _lib, external_compression as xc, records as rec

def main():
    reads_n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = "cuda:0"
    eng = xc.Engine(0)
    w = bench.build_workload(dev, reads_n, reads_n * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads_n))
    st = bench.Step(dev, eng, w)
    st.run(False)
    torch.cuda.synchronize()
    dec = bench.DecodeStep(dev, eng, w, st)
    dec.prepare()
    dec.run(False)
    torch.cuda.synchronize()
    rd = dec.reader
    names = {v: k for k, v in _lib.SERIES_ID.items()}
    full = dec.r_len.clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def t(label):
        ms = []
        for _ in range(3):
            ev[0].record()
            rd.read(dec.arena_base, dec.r_off, dec.r_len, dec.cont_first, dec.series_bytes, multiref=dec.multiref, ap_base=dec.ap_base)
            ev[1].record()
            torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1]))
        print("%-6s mean len %9.0f  ms %s" % (label, float(dec.r_len.float().sum().item()) / w["n_cont"], ["%.3f" % x for x in ms]), flush=True)
    t("all")
    for k in range(30):
        if int(full.view(-1, 30)[:, k].sum().item()) == 0:
            continue
        dec.r_len.zero_()
        dec.r_len.view(-1, 30)[:, k] = full.view(-1, 30)[:, k]
        t(names[k])
    dec.r_len.zero_()
    t("none")
    eng.close()

main()
