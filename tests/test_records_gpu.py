"""Parity of the HIP record kernels (K3 feature extraction, K4 base / quality restoration, K5 lossy quality), called
through the C ABI, with the golden vectors and with the CPU oracle on random containers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def rk():
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    eng = xc.Engine(0)
    yield records, eng
    eng.close()


def _feat_tuple(op, pos, a, b, bases):
    op = chr(op)
    if op == "X":
        return [op, pos, chr((a >> 8) & 0xFF), chr(a & 0xFF)]
    if op in "DNHP":
        return [op, pos, a]
    if op == "S":
        return [op, pos, bases[a:a + b].decode()]
    if op == "i":
        return [op, pos, chr(a)]
    if op == "Q":
        return [op, pos, a]
    raise AssertionError(op)


def _feats_of_read(f, r, bases):
    s, e = int(f["off"][r]), int(f["off"][r + 1])
    return [_feat_tuple(f["op"][i], int(f["pos"][i]), int(f["a"][i]), int(f["b"][i]), bases) for i in range(s, e)]


def _dref(ref):
    import torch
    return torch.as_tensor(np.frombuffer(bytes(ref) + b"\0" * 16, np.uint8).copy()).to(DEV)


def test_kat2_features_matrix_restore(katrec, rk):
    records, eng = rk
    ref = katrec["ref"].encode()
    d_ref = _dref(ref)
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(d_ref)
    norm = records.CramNormalizer(eng)
    for row in katrec["kat2"]:
        b = row["bases"].encode()
        reads = records.ReadBatch.from_lists(DEV, [row["start"]], [row["cigar"]], [b], ref_len=len(ref))
        feats, hist = fac.create_variations(reads)
        f = feats.to_host()
        assert _feats_of_read(f, 0, b) == row["features"], row
        m = records.substitution_matrix(hist[0])
        assert bytes(m).hex() == row["matrix"], row
        fac.assign_codes(reads, feats, m)
        out, off, st = norm.restore_read_bases(DEV, [len(b)], [row["start"]], [0], feats, d_ref, [(0, len(ref), 0)], m,
                                               seq=reads.bases, seq_off=reads.base_off)
        assert int(st[0]) == 0
        assert out[:len(b)].cpu().numpy().tobytes().decode() == row["restored"], row


def test_kat4_window_edges(katrec, rk):
    records, eng = rk
    ref = katrec["ref"].encode()
    d_ref = _dref(ref)
    norm = records.CramNormalizer(eng)
    default = np.frombuffer(bytes.fromhex("1b1b1b1b1b"), np.uint8)
    for row in katrec["kat4"]:
        lo, hi = row["window"]
        fl = row["features"]
        feats = records.FeatureBatch.from_host(
            DEV, np.array([ord(x[0]) for x in fl], np.uint8), np.array([x[1] for x in fl], np.int32),
            np.array([(x[2] << 16) if x[0] == "X" else x[2] for x in fl], np.int32), np.zeros(len(fl), np.int32),
            np.array([0, len(fl)], np.int64))
        out, off, st = norm.restore_read_bases(DEV, [row["len"]], [row["start"]], [0], feats, d_ref, [(lo, hi - lo, lo)],
                                               default)
        assert int(st[0]) == 0
        assert out[:row["len"]].cpu().numpy().tobytes().decode() == row["restored"], row


def test_kat3_lossy(katrec, rk):
    records, eng = rk
    ref = katrec["ref"].encode()
    d_ref = _dref(ref)
    k = katrec["kat3"]
    b = k["bases"].encode()
    reads = records.ReadBatch.from_lists(DEV, [k["start"]], [k["cigar"]], [b], quals=[bytes(k["quals"])], mapq=[k["mapq"]],
                                         ref_len=len(ref))
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(d_ref)
    feats, _ = fac.create_variations(reads)
    assert _feats_of_read(feats.to_host(), 0, b) == k["base_features"]
    norm = records.CramNormalizer(eng)
    span = [sum(int(c) >> 4 for c in records.pack_cigar(k["cigar"]) if (c & 15) in (0, 2, 3, 7, 8)) - 1]
    for row in k["rows"]:
        qsp = records.QualityScorePreservation(eng, row["spec"])
        wpos = qsp.tracks_windows([k["start"]], span, [0], [k["mapq"]], len(ref))
        scores, force, status = qsp.add_quality_scores(reads, d_ref, wpos)
        assert int(status[0]) == 0
        assert int(force[0]) == row["cf"], row
        merged = qsp.merge_quality_features(reads, scores, force, status, feats)
        rflags = [records.RF_FORCE_QS if int(force[0]) else 0]
        restored, st = norm.restore_quality_scores(DEV, [len(b)], rflags, merged, scores[:len(b)])
        got = [] if int(st[0]) == 1 else restored[:len(b)].cpu().numpy().tolist()
        assert got == row["restored"], row
        if "merged_prefix" in row:
            mf = _feats_of_read(merged.to_host(), 0, b)
            for want, have in zip(row["merged_prefix"], mf):
                assert have[:len(want)] == want, (want, have)


def _random_container(rng, n, ref_len, long_reads=False):
    ref = rng.choice(np.frombuffer(b"ACGTACGTACGTACGTNacgtR", np.uint8), ref_len)
    ops_pool = [0, 0, 0, 0, 1, 2, 4, 3, 5, 6, 7, 8]
    starts = np.sort(rng.integers(1, ref_len + 10, n))
    cigars, bases, quals = [], [], []
    for r in range(n):
        nel = rng.integers(20, 60) if long_reads else (1 if rng.random() < 0.7 else rng.integers(1, 9))
        ops = [(int(rng.choice(ops_pool)) if nel > 1 else 0, int(rng.integers(1, 200 if long_reads else 90)))
               for _ in range(nel)]
        L = sum(ln for op, ln in ops if op in (0, 1, 4, 7, 8))
        if L == 0:
            ops.append((0, 5))
            L = 5
        cigars.append([(ln << 4) | op for op, ln in ops])
        rp = int(starts[r]) - 1
        b = bytearray()
        for op, ln in ops:
            if op in (0, 1, 4, 7, 8):
                for i in range(ln):
                    src = ref[rp + i] if (op in (0, 7, 8) and rp + i < ref_len) else rng.choice(np.frombuffer(b"ACGT", np.uint8))
                    if rng.random() < 0.06:
                        src = rng.choice(np.frombuffer(b"ACGTNacgn", np.uint8))
                    b.append(int(src))
            if op in (0, 2, 3, 7, 8):
                rp += ln
        bases.append(bytes(b))
        quals.append(bytes(rng.integers(0, 60, L, dtype=np.uint8)))
    rflags = rng.choice(np.array([0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 32], np.uint8), n)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    st = starts.copy()
    st[rng.integers(0, n, max(1, n // 30))] = 0
    return ref, st, cigars, bases, quals, rflags, mapq


@pytest.mark.parametrize("long_reads", [False, True])
def test_random_containers_vs_oracle(rk, orc, long_reads):
    records, eng = rk
    rng = np.random.default_rng(11 + long_reads)
    n = 12 if long_reads else 300
    ref, st, cigars, bases, quals, rflags, mapq = _random_container(rng, n, 20000 if long_reads else 6000, long_reads)
    o_reads = orc.Reads(st, cigars, bases, quals=quals, mapq=mapq, rflags=rflags)
    reads = records.ReadBatch.from_lists(DEV, st, cigars, bases, quals=quals, mapq=mapq, rflags=rflags, ref_len=len(ref))
    d_ref = _dref(ref.tobytes())
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(d_ref)
    feats, hist = fac.create_variations(reads, feature_cap=16)  # forces the capacity-retry path
    want = orc.extract_features(o_reads, ref)
    got = feats.to_host()
    for key in ("op", "pos", "a", "b", "off"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(hist[0], want["hist"])
    m = records.substitution_matrix(hist[0])
    assert np.array_equal(m, orc.substitution_matrix(want["hist"]))
    fac.assign_codes(reads, feats, m)
    orc.assign_codes(want, m)
    assert np.array_equal(feats.to_host()["a"], want["a"])
    norm = records.CramNormalizer(eng)
    lens = [len(b) for b in bases]
    for lo, hi in ((0, len(ref)), (53, len(ref) - 77)):
        out, off, status = norm.restore_read_bases(DEV, lens, st, rflags, feats, d_ref, [(lo, hi - lo, lo)], m,
                                                   seq=reads.bases, seq_off=reads.base_off)
        wout, woff, wst = orc.restore_bases(lens, st, rflags, want, o_reads.bases, o_reads.base_off, ref[lo:hi], lo, m)
        wst = np.where(wst == -1, -9, wst)
        assert np.array_equal(status.cpu().numpy(), wst)
        o = out.cpu().numpy()
        for r in range(n):
            if wst[r] == 0:
                assert np.array_equal(o[woff[r]:woff[r] + lens[r]], wout[woff[r]:woff[r] + lens[r]]), r
    # pile-up terms that depend on the (unbuilt) coverage / mismatch tracks are refused, not answered with zero tracks
    for spec in ("U8-P3", "R40-X5"):
        qsp = records.QualityScorePreservation(eng, spec)
        with pytest.raises(records.CramGpuError) as ei:
            qsp.tracks_windows(st, o_reads.aln_span, rflags, mapq, len(ref), 1000000)
        assert ei.value.code == -10
        with pytest.raises(records.CramGpuError) as ei:
            qsp.add_quality_scores(reads, d_ref, np.ones(n, np.int32), 1000000)
        assert ei.value.code == -10
    for spec in ("N40-D8", "R8-N40-D8", "*8", "m5", "I40", "R8", "U8-P9999", "R40-X0"):
        qsp = records.QualityScorePreservation(eng, spec)
        pols = orc.parse_policies(spec)
        for window in (1000000, 700):
            wpos = qsp.tracks_windows(st, o_reads.aln_span, rflags, mapq, len(ref), window)
            assert np.array_equal(wpos, orc.tracks_windows(o_reads, pols, len(ref), window))
            scores, force, status = qsp.add_quality_scores(reads, d_ref, wpos, window)
            ws, wf, wst = orc.lossy_quals(o_reads, ref, pols, window, wpos)
            wst = np.where(wst == -1, -9, wst)
            assert np.array_equal(status.cpu().numpy(), wst), spec
            s = scores.cpu().numpy()
            f = force.cpu().numpy()
            for r in range(n):
                if wst[r] == 0:
                    o0 = int(o_reads.base_off[r])
                    assert f[r] == wf[r], (spec, r)
                    assert np.array_equal(s[o0:o0 + lens[r]], ws[o0:o0 + lens[r]]), (spec, r)


def test_multi_container_batch(rk, orc):
    """Two containers with their own reference windows in one call give what two single calls give."""
    records, eng = rk
    import torch
    rng = np.random.default_rng(5)
    parts = [_random_container(rng, 40, 3000), _random_container(rng, 50, 4000)]
    refs = [p[0] for p in parts]
    arena = np.concatenate([refs[0], np.zeros(16, np.uint8), refs[1], np.zeros(16, np.uint8)])
    wins = [(0, len(refs[0]), 0), (len(refs[0]) + 16, len(refs[1]), 0)]
    st = np.concatenate([p[1] for p in parts])
    cigars = parts[0][2] + parts[1][2]
    bases = parts[0][3] + parts[1][3]
    rflags = np.concatenate([p[5] for p in parts])
    cont = np.concatenate([np.zeros(40, np.int32), np.ones(50, np.int32)])
    reads = records.ReadBatch.from_lists(DEV, st, cigars, bases, rflags=rflags)
    reads.read_cont = torch.as_tensor(cont).to(DEV)
    reads.n_cont = 2
    reads.win = torch.as_tensor(np.asarray(wins, np.int64).reshape(-1)).to(DEV)
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(torch.as_tensor(arena).to(DEV))
    feats, hist = fac.create_variations(reads)
    got = feats.to_host()
    base = 0
    fbase = 0
    for c, p in enumerate(parts):
        o_reads = orc.Reads(p[1], p[2], p[3], rflags=p[5])
        want = orc.extract_features(o_reads, p[0])
        k = want["op"].size
        assert np.array_equal(got["op"][fbase:fbase + k], want["op"])
        assert np.array_equal(got["pos"][fbase:fbase + k], want["pos"])
        assert np.array_equal(got["a"][fbase:fbase + k], want["a"])
        assert np.array_equal(hist[c], want["hist"])
        fbase += k
        base += o_reads.n
    assert fbase == feats.count


def test_device_matrices_match_host_path(rk, orc):
    """cramgpu_matrices_assign_codes_dev (F3 on the device) gives the matrices and codes of the host path."""
    records, eng = rk
    import torch
    rng = np.random.default_rng(21)
    parts = [_random_container(rng, 60, 3000), _random_container(rng, 70, 2500), _random_container(rng, 5, 500)]
    refs = [p[0] for p in parts]
    arena, wins, o = [], [], 0
    for r in refs:
        arena += [r, np.zeros(16, np.uint8)]
        wins.append((o, len(r), 0))
        o += len(r) + 16
    st = np.concatenate([p[1] for p in parts])
    cigars = sum((p[2] for p in parts), [])
    bases = sum((p[3] for p in parts), [])
    rflags = np.concatenate([p[5] for p in parts])
    cont = np.concatenate([np.full(len(p[1]), c, np.int32) for c, p in enumerate(parts)])
    reads = records.ReadBatch.from_lists(DEV, st, cigars, bases, rflags=rflags)
    reads.read_cont = torch.as_tensor(cont).to(DEV)
    reads.n_cont = len(parts)
    reads.win = torch.as_tensor(np.asarray(wins, np.int64).reshape(-1)).to(DEV)
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(torch.as_tensor(np.concatenate(arena)).to(DEV))
    feats, hist = fac.create_variations(reads)
    mats = np.stack([records.substitution_matrix(hist[c]) for c in range(len(parts))])
    for c in range(len(parts)):
        assert np.array_equal(mats[c], orc.substitution_matrix(hist[c]))
    a_before = feats.a.clone()
    fac.assign_codes(reads, feats, mats)
    want_a = feats.a.clone()
    feats.a.copy_(a_before)
    d_hist = torch.as_tensor(hist.reshape(-1)).to(DEV)
    d_mats = torch.zeros(len(parts) * 5, dtype=torch.uint8, device=DEV)
    fac.matrices_and_codes(reads, feats, d_hist, d_mats)
    torch.cuda.synchronize()
    assert np.array_equal(d_mats.cpu().numpy().reshape(-1, 5), mats)
    assert torch.equal(feats.a[:feats.count], want_a[:feats.count])


def test_many_reads_offsets(rk, orc):
    """More reads than one scan tile (4096): feature offsets and features still equal the oracle's."""
    records, eng = rk
    rng = np.random.default_rng(33)
    n, ref_len = 9500, 40000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), ref_len)
    starts = np.sort(rng.integers(1, ref_len - 60, n))
    cigars, bases = [], []
    for r in range(n):
        L = int(rng.integers(20, 50))
        b = ref[starts[r] - 1:starts[r] - 1 + L].copy()
        flip = rng.random(L) < 0.08
        b[flip] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(flip.sum()))
        cigars.append([(L << 4) | 0])
        bases.append(b.tobytes())
    o_reads = orc.Reads(starts, cigars, bases)
    reads = records.ReadBatch.from_lists(DEV, starts, cigars, bases, ref_len=len(ref))
    fac = records.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(_dref(ref.tobytes()))
    feats, hist = fac.create_variations(reads)
    want = orc.extract_features(o_reads, ref)
    got = feats.to_host()
    for key in ("off", "op", "pos", "a"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(hist[0], want["hist"])


def test_restore_quals_many_reads(rk, orc):
    """restoreQualityScores over a few hundred reads: back-to-back force-preserved arrays (the wave's coalesced span
    scan), holes of -1 inside some of them, one array that is all holes (NULL_QUALS), a non-forced read with Q features
    in the middle of the run, and a tail of fewer than 64 reads."""
    records, eng = rk
    rng = np.random.default_rng(33)
    n = 64 * 5 + 23
    lens = rng.integers(100, 152, n).astype(np.int32)
    lens[:128] = 150
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    scores = rng.integers(2, 42, int(lens.sum()), dtype=np.uint8)
    rflags = np.full(n, 16, np.uint8)  # CRAMGPU_RF_FORCE_QS
    for r in (70, 71, 130, 200, 330):
        k = rng.integers(0, lens[r], 5)
        scores[off[r] + k] = 0xFF
    scores[off[131]:off[131] + lens[131]] = 0xFF
    rflags[260] = 0  # not forced: default + Q features
    f_off = np.zeros(n + 1, np.int64)
    f_off[261:] = 2
    feats = dict(op=np.array([ord("Q"), ord("Q")], np.uint8), pos=np.array([3, 7], np.int32), a=np.array([11, 12], np.int32),
                 b=np.zeros(2, np.int32), off=f_off)
    want, wst = orc.restore_quals(lens, rflags, feats, scores)
    fb = records.FeatureBatch.from_host(DEV, feats["op"], feats["pos"], feats["a"], feats["b"], feats["off"])
    norm = records.CramNormalizer(eng)
    got, st = norm.restore_quality_scores(DEV, lens, rflags, fb, np.concatenate([scores, np.zeros(16, np.uint8)]))
    assert np.array_equal(st.cpu().numpy(), wst)
    g = got.cpu().numpy()[:scores.size]
    for r in range(n):
        if wst[r] == 0:
            assert np.array_equal(g[off[r]:off[r] + lens[r]], want[off[r]:off[r] + lens[r]]), r


def test_malformed_feature_positions(rk):
    """Decode-side features come from a file: a Q / B position outside the read, or an S / I sequence outside the
    sequence arena, makes the reference throw (CramNormalizer.java:302,307: scores[pos - 1]); the kernels report
    CRAMGPU_E_JAVA_THROWS for that read and write nothing outside it."""
    records, eng = rk
    import torch
    norm = records.CramNormalizer(eng)
    for L in (20, 1500):  # the lane-per-read and the wave-per-read kernels
        lens = np.array([L, L, L, L], np.int32)
        rflags = np.zeros(4, np.uint8)  # not force-preserved: default score + Q / B patches
        f_off = np.array([0, 1, 2, 3, 4], np.int64)
        feats = dict(op=np.array([ord("Q"), ord("Q"), ord("B"), ord("Q")], np.uint8),
                     pos=np.array([0, L + 1, L + 7, L], np.int32), a=np.array([9, 9, 65, 9], np.int32),
                     b=np.array([0, 0, 9, 0], np.int32), off=f_off)
        fb = records.FeatureBatch.from_host(DEV, feats["op"], feats["pos"], feats["a"], feats["b"], feats["off"])
        guard = np.full(4 * L + 64, 0x55, np.uint8)
        got, st = norm.restore_quality_scores(DEV, lens, rflags, fb, guard)
        assert st.cpu().numpy().tolist() == [-9, -9, -9, 0]
        g = got.cpu().numpy()
        assert (g[4 * L:] == 0x55).all()  # nothing written behind the last read
        assert g[4 * L - 1] == 9 and (g[3 * L:4 * L - 1] == 30).all()
        # restoreReadBases: S feature whose bytes lie outside the sequence arena, I feature with a negative offset
        ref = np.frombuffer(b"ACGT" * 1000, np.uint8)
        d_ref = _dref(ref.tobytes())
        feats = dict(op=np.array([ord("S"), ord("I"), ord("S")], np.uint8), pos=np.array([1, 2, 1], np.int32),
                     a=np.array([10 ** 6, -5, 0], np.int32), b=np.array([4, 3, 4], np.int32),
                     off=np.array([0, 1, 2, 3, 3], np.int64))
        fb = records.FeatureBatch.from_host(DEV, feats["op"], feats["pos"], feats["a"], feats["b"], feats["off"])
        seq = np.frombuffer(b"TTTTGGGGCCCCAAAA", np.uint8)
        out, off, st = norm.restore_read_bases(DEV, lens, np.array([1, 1, 1, 1], np.int32), rflags, fb, d_ref,
                                               [(0, ref.size, 0)], np.frombuffer(bytes.fromhex("1b1b1b1b1b"), np.uint8),
                                               seq=seq, seq_off=np.zeros(4, np.int64))
        assert st.cpu().numpy().tolist() == [-9, -9, 0, 0]
        o = out.cpu().numpy()
        assert o[2 * L:2 * L + 8].tobytes() == b"TTTTACGT"
