#!/usr/bin/env python
"""Times the record kernels of the bench step (K3 two passes + scan, matrices/codes) and of the decode leg."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
import bench
from cramtools_amd import _lib, external_compression as xc

def main():
    reads = 10_000_000
    dev = "cuda:0"
    eng = xc.Engine(0)
    w = bench.build_workload(dev, reads, reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads))
    st = bench.Step(dev, eng, w)
    L = _lib.lib()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(4):
        st.hist.zero_()
        rs_, fs_ = w["reads"].struct(), st.feats.struct()
        ev[0].record()
        n = _lib.check(L.cramgpu_extract_features_dev(eng.handle, C.byref(rs_), w["ref"].data_ptr(), C.byref(fs_), st.hist.data_ptr(), s))
        ev[1].record()
        st.feats.count = int(n)
        st.fac.matrices_and_codes(w["reads"], st.feats, st.hist, st.mats)
        ev[2].record()
        torch.cuda.synchronize()
        print("K3 %.3f ms  matrices+codes %.3f ms  features %d" % (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), n), flush=True)
    st.run(False)
    dec = bench.DecodeStep(dev, eng, w, st)
    dec.prepare()
    for rep in range(3):
        dec.run(True)
    print("decode leg ms", ["%.2f" % m for m in dec.ms], "K2", ["%.2f" % m for m in dec.k2_ms], flush=True)
    eng.close()

if __name__ == "__main__":
    main()
