"""The bench's encode step and decode leg on the GPU at 1/50 of configs[1] (200 000 reads, 20 containers): every
encoded block equals the oracle's stream, the features of the lane-per-read K3 kernel equal the oracle's, and the
decode leg (rANS decode + restoreReadBases + restoreQualityScores) round-trips blocks and bases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bench_step_and_decode_leg(orc):
    import torch
    import bench
    from cramtools_amd import external_compression as xc
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = "cuda:0"
    eng = xc.Engine(0)
    try:
        n_reads = 200_000
        w = bench.build_workload(dev, n_reads, 1_280_000, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, n_reads))
        step = bench.Step(dev, eng, w)
        step.run(True)
        lens = step.d_len.cpu().numpy()
        pos = step.d_pos.cpu().numpy()
        d_in = w["d_in"].cpu().numpy()
        d_out = step.d_out.cpu().numpy()
        assert (lens >= 0).all()
        for i in range(w["n_blocks"]):
            b = w["blocks"][i]
            want = orc.rans(d_in[b.in_off:b.in_off + b.in_len].tobytes(), b.order)
            assert d_out[int(pos[i]):int(pos[i]) + int(lens[i])].tobytes() == want, i
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
        r.cigar = np.ascontiguousarray(reads.cigar.cpu().numpy().view(np.uint32))
        want = orc.extract_features(r, w["ref"].cpu().numpy()[:-16])
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
