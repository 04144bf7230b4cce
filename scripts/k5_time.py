#!/usr/bin/env python
"""K5 (lossy quality model) and the merge of its Q features on the configs[1] reads under R8-N40-D8: kernel times with
events on the launch stream.  The windows of ReferenceTracks are replayed on the host first (cramgpu_tracks_windows)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cramtools_amd import _lib, external_compression as xc, records as rec

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    spec = sys.argv[2] if len(sys.argv) > 2 else "R8-N40-D8"
    dev = "cuda:0"
    eng = xc.Engine(0)
    w = bench.build_workload(dev, n_reads, n_reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, n_reads))
    reads, rs = w["reads"], w["rs"]
    st = bench.Step(dev, eng, w, writer=False)
    st.run(False)
    torch.cuda.synchronize()
    qsp = rec.QualityScorePreservation(eng, spec)
    t0 = time.perf_counter()
    wpos = qsp.tracks_windows(rs.aln_start.cpu().numpy(), rs.aln_span.cpu().numpy(), rs.rflags.cpu().numpy(), rs.mapq.cpu().numpy(),
                              n_reads * 64 // 10)
    t_host = time.perf_counter() - t0
    d_wpos = torch.as_tensor(wpos).to(dev)
    scores = torch.zeros(reads.bases.numel(), dtype=torch.uint8, device=dev)
    force = torch.zeros(reads.n, dtype=torch.uint8, device=dev)
    status = torch.zeros(reads.n, dtype=torch.int32, device=dev)
    out = rec.FeatureBatch(dev, reads.n, st.feats.count + int(reads.bases.numel()) // 4 + 16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    L, s = _lib.lib(), torch.cuda.current_stream().cuda_stream
    rs_, fi, fo = reads.struct(), st.feats.struct(), out.struct()
    for rep in range(3):
        ev[0].record()
        _lib.check(L.cramgpu_lossy_quals_dev(eng.handle, C.byref(rs_), w["ref"].data_ptr(), d_wpos.data_ptr(), qsp.WINDOW, qsp.policies,
                                             qsp.n_pol, scores.data_ptr(), force.data_ptr(), status.data_ptr(), s))
        ev[1].record()
        n = _lib.check(L.cramgpu_merge_quality_features_dev(eng.handle, reads.n, reads.read_len.data_ptr(), reads.base_off.data_ptr(),
                                                            scores.data_ptr(), force.data_ptr(), status.data_ptr(), C.byref(fi),
                                                            C.byref(fo), s))
        ev[2].record()
        torch.cuda.synchronize()
        tot = int(rs.read_len.sum().item())
        print("%s on %d reads (%d bases): K5 %.3f ms (%.0f GB/s of 2 x scores + bases + span), merge %.3f ms; features %d -> %d, "
              "forced reads %d, status != 0: %d; host window replay %.1f ms" % (
                  spec, reads.n, tot, ev[0].elapsed_time(ev[1]), (3 * tot + n_reads * 64 // 10) / ev[0].elapsed_time(ev[1]) / 1e6,
                  ev[1].elapsed_time(ev[2]), st.feats.count, int(n), int(force.sum().item()), int((status != 0).sum().item()),
                  t_host * 1e3), flush=True)
    eng.close()

main()
