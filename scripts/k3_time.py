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
    import time
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for mode in ("K1 alone", "memset then K1", "K3 then K1", "memset, page-touch both, K1"):
        ms = []
        for rep in range(3):
            if "K3" in mode:
                st.hist.zero_()
                n = _lib.check(L.cramgpu_extract_features_dev(eng.handle, C.byref(rs_), w["ref"].data_ptr(), C.byref(fs_), st.hist.data_ptr(), s))
                st.fac.matrices_and_codes(w["reads"], st.feats, st.hist, st.mats)
            if "matrices" in mode:
                st.fac.matrices_and_codes(w["reads"], st.feats, st.hist, st.mats)
            if "sleep" in mode:
                torch.cuda.synchronize()
                time.sleep(0.02)
            if "memset" in mode:
                big.zero_()
            if "page-touch d_out" in mode or "page-touch both" in mode:
                junk = st.d_out[::4096].sum()
            if "page-touch d_in" in mode or "page-touch both" in mode:
                junk2 = w["d_in"][::4096].sum()
            if "2M-touch" in mode:
                junk = st.d_out[::1 << 21].sum() + w["d_in"][::1 << 21].sum()
            if "gather" in mode:
                eng.gather_dev(st.d_out, st.d_pos, st.d_len, w["n_blocks"], w["max_cap"], st.d_pack, st.d_pack_off)
            if "read d_in" in mode:
                junk = w["d_in"][:1 << 28].sum()
            if "read d_pack" in mode:
                junk = st.d_pack[:220 << 20].sum()
            ev[0].record()
            eng.encode_dev(w["d_in"], st.d_out, w["blocks"], w["n_blocks"], st.d_len, st.d_pos)
            ev[1].record()
            eng.gather_dev(st.d_out, st.d_pos, st.d_len, w["n_blocks"], w["max_cap"], st.d_pack, st.d_pack_off)
            torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1]))
        print("%-28s K1 ms %s" % (mode, ["%.2f" % m for m in ms]), flush=True)
    st.run(False)
    dec = bench.DecodeStep(dev, eng, w, st)
    dec.prepare()
    for rep in range(3):
        dec.run(True)
    print("decode leg ms", ["%.2f" % m for m in dec.ms], "K2", ["%.2f" % m for m in dec.k2_ms], flush=True)
    eng.close()

if __name__ == "__main__":
    main()
