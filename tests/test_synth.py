"""The torch (device) generators of the bench workload against the numpy ones that the golden vectors pin, and the
vectorised read model against the oracle (CPU only)."""
import numpy as np
import torch

from cramtools_amd import synth, synth_torch


def test_lcg_and_quals_match_numpy():
    for n, seed in ((1, 1), (7, 3), (1000, 2), (70001, 0xC0FFEE)):
        assert np.array_equal(synth_torch.lcg_stream(n, seed).numpy(), synth.lcg_stream(n, seed).astype(np.int64))
        assert np.array_equal(synth_torch.lcg_bytes(n, seed, b"ACGT").numpy(), synth.lcg_bytes(n, seed, b"ACGT"))
    assert np.array_equal(synth_torch.markov_quals(65536, 2).numpy(), synth.markov_quals(65536, 2))
    # chunked generation carries the chain across chunk borders
    assert np.array_equal(synth_torch.markov_quals(5000, 2, chunk=777).numpy(), synth.markov_quals(5000, 2))
    assert synth_torch.lcg_state_after(12345, 9) == int(synth.lcg_stream(12345, 9)[-1])


def test_itf8_matches_numpy():
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.integers(0, 1 << k, 200) for k in (6, 8, 13, 15, 20, 22, 27, 29, 32)] + [[-1, 0, 127, 128]])
    b, off = synth_torch.itf8(torch.as_tensor(v))
    assert np.array_equal(b.numpy(), synth.itf8(v))
    assert off[-1].item() == b.numel()


def test_read_model_round_trips_through_the_oracle(orc):
    ref = synth_torch.reference(200000)
    rs = synth_torch.make_reads(3000, ref, indel_rate=0.002, softclip_frac=0.05, sub_rate=0.01, chunk_reads=1000)
    L = 150
    assert rs.bases.numel() == 3000 * L + 16
    st = rs.aln_start.numpy()
    assert (np.diff(st) >= 0).all() and st.min() >= 1
    nc = rs.n_cigar.numpy()
    assert set(np.unique(nc)) <= {1, 2, 3, 4} and (nc > 1).sum() > 50
    # every CIGAR consumes exactly L read bases
    cig = rs.cigar.numpy().view(np.uint32)
    coff = rs.cigar_off.numpy()
    for r in range(0, 3000, 37):
        ops = cig[coff[r]:coff[r] + nc[r]]
        assert sum(int(c) >> 4 for c in ops if (int(c) & 15) in (0, 1, 4, 7, 8)) == L

    class R:
        pass
    reads = R()
    reads.n = rs.n
    for k in ("bases", "base_off", "read_len", "aln_start", "rflags", "cigar_off", "n_cigar"):
        setattr(reads, k, np.ascontiguousarray(getattr(rs, k).numpy()))
    reads.cigar = np.ascontiguousarray(cig)
    refb = ref.numpy()
    f = orc.extract_features(reads, refb)
    # features: substitutions at ~1 %, plus clips and indels
    nx = int((f["op"] == ord("X")).sum())
    assert 0.005 * 3000 * L < nx < 0.02 * 3000 * L
    m = orc.substitution_matrix(f["hist"])
    orc.assign_codes(f, m)
    out, off, status = orc.restore_bases(reads.read_len, reads.aln_start, reads.rflags, f, reads.bases, reads.base_off,
                                         refb, 0, m)
    assert (status == 0).all()
    assert np.array_equal(out, reads.bases[:3000 * L])
