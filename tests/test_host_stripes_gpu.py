"""Level 2 through HOST arrays only (numpy + ctypes, no torch tensor, no HIP call on this side): the call sequence of
the JNI shim — cramgpu_ref_upload, cramgpu_submit_reads (K3 -> F3 -> K5 -> merge), cramgpu_submit_restore (K4 + N2),
cramgpu_job_wait — against the oracle on random containers over two contigs."""
import numpy as np
import pytest

from test_records_gpu import _random_container

pytestmark = pytest.mark.gpu


def _two_containers(orc, seed):
    rng = np.random.default_rng(seed)
    parts = [_random_container(rng, 60, 5000), _random_container(rng, 45, 3500)]
    cat = lambda k: sum((list(p[k]) for p in parts), [])  # noqa: E731
    starts, cigars, bases, quals = cat(1), cat(2), cat(3), cat(4)
    rflags = np.concatenate([p[5] for p in parts])
    mapq = np.concatenate([p[6] for p in parts])
    o_parts = [orc.Reads(p[1], p[2], p[3], quals=p[4], mapq=p[6], rflags=p[5]) for p in parts]
    return parts, o_parts, starts, cigars, bases, quals, rflags, mapq


def test_host_stripes_encode_and_restore(orc):
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    eng = xc.Engine(0)
    try:
        parts, o_parts, starts, cigars, bases, quals, rflags, mapq = _two_containers(orc, 21)
        for k, p in enumerate(parts):
            records.ref_upload(eng, 3 + k, p[0])  # contig ids need not start at 0
        cont_reads, cont_contig = [len(p[1]) for p in parts], [3, 4]
        # capacity too small first: the job says how much it needs
        job = records.HostReadJob(eng, starts, cigars, bases, cont_reads, cont_contig, rflags=rflags, feat_cap=8)
        with pytest.raises(records.CramGpuError) as ei:
            job.wait()
        assert ei.value.code == -3
        need = int(job.f_off[job.n])
        got = records.HostReadJob(eng, starts, cigars, bases, cont_reads, cont_contig, rflags=rflags, feat_cap=need).wait()
        wants = [orc.extract_features(o, p[0]) for o, p in zip(o_parts, parts)]
        f0 = 0
        r0 = 0
        for c, (w, o) in enumerate(zip(wants, o_parts)):
            k = len(w["op"])
            m = orc.substitution_matrix(w["hist"])
            orc.assign_codes(w, m)
            assert np.array_equal(got["matrices"][c], m)
            assert np.array_equal(got["hist"][c], w["hist"])
            assert np.array_equal(got["off"][r0:r0 + o.n + 1] - f0, w["off"])
            for key in ("op", "pos", "a", "b"):
                assert np.array_equal(got[key][f0:f0 + k], w[key]), (c, key)
            f0 += k
            r0 += o.n
        assert f0 == len(got["op"]) == need
        # decode side: restoreReadBases through ReferenceRegion windows + restoreQualityScores on force-preserved scores
        lens = [len(b) for b in bases]
        base_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        seq = np.frombuffer(b"".join(bases) + b"\0" * 16, np.uint8)
        wins = [(40, len(parts[0][0]) - 90), (0, len(parts[1][0]))]
        sc_in = np.frombuffer(b"".join(quals), np.uint8).copy()
        sc_in[5::17] = 0xFF  # holes that become the default score
        rj = records.HostRestoreJob(eng, lens, starts, rflags | 16, cont_reads, cont_contig, got, got["matrices"], seq=seq,
                                    seq_off=base_off, scores=sc_in, windows=wins)
        b_out, st_b, s_out, st_q = rj.wait()
        r0 = f0 = 0
        for c, (w, o, p) in enumerate(zip(wants, o_parts, parts)):
            lo, ln = wins[c]
            wout, woff, wst = orc.restore_bases(o.read_len, o.aln_start, o.rflags | 16, w, o.bases, o.base_off, p[0][lo:lo + ln],
                                                lo, got["matrices"][c])
            wst = np.where(wst == -1, -9, wst)
            assert np.array_equal(st_b[r0:r0 + o.n], wst), c
            bo = int(base_off[r0])
            for r in range(o.n):
                if wst[r] == 0:
                    a = bo + int(woff[r])
                    assert np.array_equal(b_out[a:a + o.read_len[r]], wout[woff[r]:woff[r] + o.read_len[r]]), (c, r)
            tot = int(o.read_len.sum())
            wq, wqst = orc.restore_quals(o.read_len, o.rflags | 16, w, sc_in[bo:bo + tot])
            assert np.array_equal(st_q[r0:r0 + o.n], wqst), c
            for r in range(o.n):
                if wqst[r] == 0:
                    a = int(o.base_off[r])
                    assert np.array_equal(s_out[bo + a:bo + a + o.read_len[r]], wq[a:a + o.read_len[r]]), (c, r)
            r0 += o.n
    finally:
        eng.close()


def test_host_stripes_lossy(orc):
    """The lossy leg of cramgpu_submit_reads: policies parsed from the spec, ReferenceTracks windows replayed per contig
    (and carried from one submission to the next), scores / force flags / merged features against the oracle."""
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    eng = xc.Engine(0)
    try:
        rng = np.random.default_rng(8)
        ref, st, cigars, bases, quals, rflags, mapq = _random_container(rng, 80, 6000)
        st = np.where(st == 0, 0, st)
        keep = np.ones(len(st), bool)  # sorted starts with some zeros (unplaced) in between: fine for ReferenceTracks
        records.ref_upload(eng, 0, ref)
        o = orc.Reads(st, cigars, bases, quals=quals, mapq=mapq, rflags=rflags)
        feats = orc.extract_features(o, ref)
        orc.assign_codes(feats, orc.substitution_matrix(feats["hist"]))
        for spec, window in (("R8-N40-D8", 0), ("N40-D8", 700), ("*8", 0), ("m5", 0)):
            records.ref_upload(eng, 0, ref)  # resets the contig's window position
            pols = orc.parse_policies(spec)
            W = window or 1000000
            rc = orc.tracks_windows_rc(o, pols, ref.size, W)
            job = records.HostReadJob(eng, st, cigars, bases, [o.n], [0], quals=quals, mapq=mapq, rflags=rflags,
                                      lossy_spec=spec, window=window, feat_cap=len(feats["op"]) + int(o.read_len.sum()) + 8)
            if rc < 0:
                with pytest.raises(records.CramGpuError) as ei:
                    job.wait()
                assert ei.value.code == -9
                continue
            got = job.wait()
            wpos = orc.tracks_windows(o, pols, ref.size, W)
            ws, wf, wst = orc.lossy_quals(o, ref, pols, W, wpos)
            wst9 = np.where(wst == -1, -9, wst)
            assert np.array_equal(got["status"], wst9), spec
            for r in range(o.n):
                if wst[r] == 0:
                    a = int(o.base_off[r])
                    assert got["force"][r] == wf[r], (spec, r)
                    assert np.array_equal(got["scores"][a:a + o.read_len[r]], ws[a:a + o.read_len[r]]), (spec, r)
            qs = np.where(wst < 0, 2, wst).astype(np.int32)  # a read the model threw on carries no Q features
            wm = orc.merge_quality_features(o, ws, wf, qs, feats)
            ok = wst >= 0
            for r in np.nonzero(ok)[0]:
                a0, a1 = int(got["off"][r]), int(got["off"][r + 1])
                b0, b1 = int(wm["off"][r]), int(wm["off"][r + 1])
                for key in ("op", "pos", "a", "b"):
                    assert np.array_equal(got[key][a0:a1], wm[key][b0:b1]), (spec, r, key)
        with pytest.raises(records.CramGpuError):
            records.HostReadJob(eng, st, cigars, bases, [o.n], [0], quals=quals, mapq=mapq, lossy_spec="R8-X5")
        del keep
    finally:
        eng.close()


@pytest.mark.parametrize("lossy", [False, True])
def test_host_stripes_capacity_zero_touches_nothing(orc, lossy):
    """with a feature capacity far below the real count the first set is never written, so nothing that
    walks its offsets (code assignment, the count pass of the quality merge) may run: the job reports
    CRAMGPU_E_INSUFFICIENT with the real count in off[n], and the same context still gives the oracle's answer afterwards
    (the stray writes used to land in the job arena's other columns)."""
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    eng = xc.Engine(0)
    try:
        rng = np.random.default_rng(31)
        ref, st, cigars, bases, quals, rflags, mapq = _random_container(rng, 300, 9000)
        records.ref_upload(eng, 0, ref)
        o = orc.Reads(st, cigars, bases, quals=quals, mapq=mapq, rflags=rflags)
        feats = orc.extract_features(o, ref)
        n_feat = len(feats["op"])
        assert n_feat > 300  # many features per read on average
        kw = dict(quals=quals, mapq=mapq, rflags=rflags)
        if lossy:
            kw.update(lossy_spec="*8", window=0)
        for cap in (0, 1, 7):
            records.ref_upload(eng, 0, ref)
            job = records.HostReadJob(eng, st, cigars, bases, [o.n], [0], feat_cap=cap, **kw)
            with pytest.raises(records.CramGpuError) as ei:
                job.wait()
            assert ei.value.code == -3
            if not lossy:
                assert int(job.f_off[job.n]) == n_feat
        records.ref_upload(eng, 0, ref)
        got = records.HostReadJob(eng, st, cigars, bases, [o.n], [0], feat_cap=n_feat + int(o.read_len.sum()) + 8, **kw).wait()
        orc.assign_codes(feats, orc.substitution_matrix(feats["hist"]))
        if lossy:
            pols = orc.parse_policies("*8")
            wpos = orc.tracks_windows(o, pols, ref.size, 1000000)
            ws, wf, wst = orc.lossy_quals(o, ref, pols, 1000000, wpos)
            want = orc.merge_quality_features(o, ws, wf, np.where(wst < 0, 2, wst).astype(np.int32), feats)
        else:
            want = feats
        k = len(want["op"])
        assert np.array_equal(got["off"], want["off"])
        for key in ("op", "pos", "a", "b"):
            assert np.array_equal(got[key][:k], want[key]), key
    finally:
        eng.close()
