"""bench.py's workload builder and CPU-baseline leg on a small CPU instance (no GPU): every block of the corpus is a
valid rANS input, the series have the expected shape, and the baseline leg's streams round-trip through the oracle."""
import numpy as np
import torch

import bench
from cramtools_amd import records as rec


def _oracle_features(orc):
    def features_of(reads, ref):
        class R:
            pass
        r = R()
        r.n = reads.n
        for k in ("bases", "base_off", "read_len", "aln_start", "rflags", "cigar_off", "n_cigar"):
            setattr(r, k, np.ascontiguousarray(getattr(reads, k).numpy()))
        r.cigar = np.ascontiguousarray(reads.cigar.numpy().view(np.uint32))
        f = orc.extract_features(r, ref.numpy()[:-16])
        return torch.as_tensor(f["op"]), torch.as_tensor(f["a"]), torch.as_tensor(f["off"])
    return features_of


def test_workload_and_cpu_baseline(orc, monkeypatch):
    monkeypatch.setattr(bench, "READS_PER_CONTAINER", 500)
    w = bench.build_workload("cpu", 2000, 50000, 150, 0, _oracle_features(orc))
    ns = len(bench.SERIES)
    assert w["n_cont"] == 4 and w["n_blocks"] == 4 * ns
    lens = w["lens"]
    names = [n for n, _ in bench.SERIES]
    assert (lens[:, names.index("QS")] == 500 * 150).all()
    assert (lens[:, names.index("RL")] == 500 * 2).all()      # ITF8(150) = 2 bytes
    assert (lens[:, names.index("RG")] == 500 * 5).all()      # ITF8(-1) = 5 bytes
    assert (lens[:, names.index("CF")] == 500).all()
    d_in = w["d_in"].numpy()
    for i in range(w["n_blocks"]):
        b = w["blocks"][i]
        assert b.in_off % 16 == 0 and b.out_off % 16 == 0 and b.order == bench.SERIES[i % ns][1]
        raw = d_in[b.in_off:b.in_off + b.in_len].tobytes()
        assert orc.unrans(orc.rans(raw, b.order)) == raw
    # QS block of container 1 is that container's slice of the quality stream
    b = w["blocks"][1 * ns + names.index("QS")]
    assert np.array_equal(d_in[b.in_off:b.in_off + b.in_len], w["rs"].quals.numpy()[500 * 150:1000 * 150])
    base, (out, ooff, olen, n_c) = bench.cpu_baseline(w, budget_s=0.5)
    assert base["kind"] == "port" and base["value"] > 0 and 1 <= n_c <= 4
    for i in range(n_c * ns):
        b = w["blocks"][i]
        raw = d_in[b.in_off:b.in_off + b.in_len].tobytes()
        assert out[ooff[i]:ooff[i] + olen[i]].tobytes() == orc.rans(raw, b.order)
