"""The bench's encode step and decode leg on the GPU at 1/50 of configs[1] (200 000 reads, 20 containers): every
encoded block equals the oracle's stream, the features of the lane-per-read K3 kernel equal the oracle's, and the
decode leg (rANS decode + restoreReadBases + restoreQualityScores) round-trips blocks and bases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_reads,ref_len,check_cont", [(200_000, 1_280_000, 20), (10_000_000, 64_000_000, 12)],
                         ids=["200k", "configs1-full-size"])
def test_bench_step_and_decode_leg(orc, n_reads, ref_len, check_cont):
    """At BASELINE.json's full size (10 M reads, 1000 containers) the blocks of the first containers are compared with
    the oracle byte by byte and everything else through the size-independent property: decode(encode(x)) == x for all
    10 000 blocks and restoreReadBases(features(reads)) == reads for all reads."""
    import torch
    import bench
    from cramtools_amd import external_compression as xc
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = "cuda:0"
    eng = xc.Engine(0)
    try:
        w = bench.build_workload(dev, n_reads, ref_len, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, n_reads))
        step = bench.Step(dev, eng, w)
        step.run(True)
        lens = step.d_len.cpu().numpy()
        pos = step.d_pos.cpu().numpy()
        assert (lens >= 0).all()
        for i in range(check_cont * len(bench.SERIES)):
            b = w["blocks"][i]
            raw = w["d_in"][b.in_off:b.in_off + b.in_len].cpu().numpy().tobytes()
            got = step.d_out[int(pos[i]):int(pos[i]) + int(lens[i])].cpu().numpy().tobytes()
            assert got == orc.rans(raw, b.order), i
        # K3 (a lane per read) against the oracle on the first two containers
        reads = w["reads"]
        n0 = 2 * bench.READS_PER_CONTAINER

        class R:
            pass
        r = R()
        r.n = n0
        for k in ("base_off", "read_len", "aln_start", "rflags", "cigar_off", "n_cigar"):
            setattr(r, k, np.ascontiguousarray(getattr(reads, k)[:n0].cpu().numpy()))
        r.bases = np.ascontiguousarray(reads.bases[:n0 * 150 + 16].cpu().numpy())
        n_cig = int(reads.cigar_off[n0 - 1].item()) + int(reads.n_cigar[n0 - 1].item())
        r.cigar = np.ascontiguousarray(reads.cigar[:n_cig + 1].cpu().numpy().view(np.uint32))
        ref_hi = min(ref_len, int(reads.aln_start[n0 - 1].item()) + 400)  # reads are sorted by start
        want = orc.extract_features(r, w["ref"][:ref_hi].cpu().numpy())
        k = int(want["off"][-1])
        off = step.feats.off[:n0 + 1].cpu().numpy()
        assert np.array_equal(off, want["off"])
        assert np.array_equal(step.feats.op[:k].cpu().numpy(), want["op"])
        assert np.array_equal(step.feats.pos[:k].cpu().numpy(), want["pos"])
        assert np.array_equal(step.feats.a[:k].cpu().numpy() & 0xFFFF, want["a"] & 0xFFFF)  # bits 16+ = codes
        # decode leg
        dec = bench.DecodeStep(dev, eng, w, step)
        dec.prepare()
        dec.run(True)
        assert "round-trip" in dec.verify()
    finally:
        eng.close()
