#!/usr/bin/env python
"""configs[4]-like run: long reads (10 kbp, 3 % substitutions, one indel each), lossy model R8-N40-D8, order-1 rANS of
the binned qualities.  Prints the time of every kernel of the path and checks K3 / K5 on a sample against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
from cramtools_amd import _lib, external_compression as xc, records as rec, synth_torch

def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    L = 10000
    per_cont = 10000
    dev = "cuda:0"
    eng = xc.Engine(0)
    ref_len = 8_000_000
    ref = torch.cat([synth_torch.reference(ref_len, device=dev), torch.zeros(16, dtype=torch.uint8, device=dev)])
    rs = synth_torch.make_reads(n_reads, ref[:ref_len], read_len=L, seed=1, sub_rate=0.03, indel_rate=0.0001, chunk_reads=2000)
    n_cont = (n_reads + per_cont - 1) // per_cont
    read_cont = (torch.arange(n_reads, device=dev) // per_cont).to(torch.int32)
    reads = rec.ReadBatch(dev, rs.bases, rs.base_off, rs.read_len, rs.aln_start, rs.cigar, rs.cigar_off, rs.n_cigar,
                          quals=rs.quals, mapq=rs.mapq, rflags=rs.rflags, read_cont=read_cont, windows=[(0, ref_len, 0)] * n_cont)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def timed(label, fn, nbytes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-34s %9.3f ms  %8.1f MB/s" % (label, dt * 1e3, nbytes / dt / 1e6), flush=True)
        return r
    nb = n_reads * L
    fac = rec.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(ref)
    fac.create_variations(reads, feature_cap=nb // 20)
    feats, hist = timed("K3 extract (wave per tile, 2 passes)", lambda: fac.create_variations(reads, feature_cap=nb // 20), nb)
    print("   features %d (%.1f per read)" % (feats.count, feats.count / n_reads))
    qsp = rec.QualityScorePreservation(eng, "R8-N40-D8")
    st_h, sp_h = rs.aln_start.cpu().numpy(), rs.aln_span.cpu().numpy()
    fl_h, mq_h = rs.rflags.cpu().numpy(), rs.mapq.cpu().numpy()
    t0 = time.perf_counter()
    wpos = qsp.tracks_windows(st_h, sp_h, fl_h, mq_h, ref_len)
    print("%-34s %9.3f ms  (host)" % ("ReferenceTracks window replay", (time.perf_counter() - t0) * 1e3))
    qsp.add_quality_scores(reads, ref, wpos)
    scores, force, status = timed("K5 lossy quality R8-N40-D8", lambda: qsp.add_quality_scores(reads, ref, wpos), 2 * nb)
    print("   force-preserved reads %d of %d" % (int(force.sum().item()), n_reads))
    merged = timed("merge Q features", lambda: qsp.merge_quality_features(reads, scores, force, status, feats), nb)
    # QS blocks: one per container (forced reads carry their whole binned array)
    lens = [min(per_cont, n_reads - c * per_cont) * L for c in range(n_cont)]
    blocks, in_bytes, out_bytes = xc.Engine.plan_encode(lens, [1] * n_cont)
    d_in = torch.zeros(in_bytes, dtype=torch.uint8, device=dev)
    o = 0
    for c in range(n_cont):
        d_in[blocks[c].in_off:blocks[c].in_off + lens[c]] = scores[o:o + lens[c]]
        o += lens[c]
    d_out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n_cont, dtype=torch.int32, device=dev)
    d_pos = torch.zeros(n_cont, dtype=torch.int64, device=dev)
    timed("K1 rANS o1 of %d x %d MB QS blocks" % (n_cont, lens[0] // 1000000), lambda: eng.encode_dev(d_in, d_out, blocks, n_cont, d_len, d_pos), nb)
    cl = d_len.cpu().numpy()
    print("   compressed %.1f MB (%.3f B/B)" % (cl.sum() / 1e6, cl.sum() / nb))
    # sample parity: first 8 reads against the oracle
    from oracle import orc
    orc.build()
    k = 8
    class R: pass
    r = R(); r.n = k
    for name in ("base_off", "read_len", "aln_start", "rflags", "cigar_off", "n_cigar"):
        setattr(r, name, np.ascontiguousarray(getattr(reads, name)[:k].cpu().numpy()))
    r.bases = np.ascontiguousarray(reads.bases[:k * L + 16].cpu().numpy())
    ncg = int(reads.cigar_off[k - 1].item()) + int(reads.n_cigar[k - 1].item())
    r.cigar = np.ascontiguousarray(reads.cigar[:ncg + 1].cpu().numpy().view(np.uint32))
    hi = min(ref_len, int(reads.aln_start[k - 1].item()) + L + 100)
    want = orc.extract_features(r, ref[:hi].cpu().numpy())
    kf = int(want["off"][-1])
    assert np.array_equal(feats.off[:k + 1].cpu().numpy(), want["off"])
    assert np.array_equal(feats.op[:kf].cpu().numpy(), want["op"]) and np.array_equal(feats.pos[:kf].cpu().numpy(), want["pos"])
    print("sample parity: features of the first %d reads equal the oracle's (%d features)" % (k, kf))
    eng.close()

if __name__ == "__main__":
    main()
