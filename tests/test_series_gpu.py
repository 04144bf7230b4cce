"""The data-series writer (cramgpu_write_series_dev) against the oracle's Writer#write
(oracle/series.c): random slices with every operator, every record kind and every ITF8 width, several containers per
batch (multi-reference and explicit slice starts included), the capacity guard, and the series of a K3-extracted
container feeding the rANS encoder as they lie."""
import numpy as np
import pytest

from tests import series_cases as sc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _batch(rng, sizes):
    """Several random slices back to back as one batch: (ReadBatch, RecordColumns kwargs, FeatureBatch, per-slice dicts)."""
    from cramtools_amd import records
    parts = [sc.random_slice(rng, n) for n in sizes]
    n = sum(sizes)
    read_len = np.concatenate([p["read_len"] for p in parts])
    base_off = np.concatenate([[0], np.cumsum(read_len)[:-1]]).astype(np.int64)
    bases = np.concatenate([p["bases"][:int(p["base_off"][-1])] for p in parts] + [np.zeros(16, np.uint8)])
    quals = np.concatenate([p["quals"][:int(p["base_off"][-1])] for p in parts] + [np.zeros(16, np.uint8)])
    aln_start = np.concatenate([p["aln_start"] for p in parts])
    mapq = np.concatenate([p["mapq"] for p in parts])
    read_cont = np.concatenate([np.full(s, c, np.int32) for c, s in enumerate(sizes)])
    reads = records.ReadBatch(DEV, bases, base_off, read_len, aln_start, np.zeros(1, np.int32), np.zeros(n, np.int64),
                              np.zeros(n, np.int32), quals=quals, mapq=mapq, read_cont=read_cont,
                              windows=[(0, 0, 0)] * len(sizes))
    cols = {k: np.concatenate([p["cols"][k] for p in parts]) for k in records.RECORD_COLS}
    f_off = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p["feats"]["off"]) for p in parts]))]).astype(np.int64)
    cat = lambda k, dt: np.concatenate([p["feats"][k] for p in parts]).astype(dt)
    feats = records.FeatureBatch.from_host(DEV, cat("op", np.uint8), cat("pos", np.int32), cat("a", np.int32), cat("b", np.int32), f_off)
    return reads, cols, feats, parts


@pytest.mark.parametrize("seed,sizes", [(0, [1]), (1, [300]), (2, [700, 5, 256, 257]), (3, [64, 1, 1000])])
def test_writer_matches_oracle(orc, seed, sizes):
    from cramtools_amd import external_compression as xc, records
    rng = np.random.default_rng(100 + seed)
    reads, cols, feats, parts = _batch(rng, sizes)
    n_cont = len(sizes)
    multiref = np.array([(c + seed) & 1 for c in range(n_cont)], np.uint8)
    ap_base = np.array([int(p["aln_start"][0]) - (c % 3) for c, p in enumerate(parts)], np.int32)
    ap_delta = seed != 3
    eng = xc.Engine(0)
    try:
        # alias_qs on for one case: no container of these slices qualifies (not every record keeps its scores), nothing changes
        rc = records.RecordColumns(DEV, reads.n, multiref=multiref, ap_base=ap_base, ap_delta=ap_delta, alias_qs=seed == 2, **cols)
        w = records.Writer(eng, DEV, n_cont, 8 * int(reads.bases.numel()) + 70 * reads.n + 4096)
        w.write(reads, rc, feats)
        off, ln = w.lengths()
        assert (off % 16 == 0).all()
        arena = w.arena.cpu().numpy()
        for c, p in enumerate(parts):
            want = orc.write_series(p["cols"], p["read_len"], p["aln_start"], p["mapq"], p["bases"], p["quals"], p["base_off"],
                                    p["feats"], multiref=bool(multiref[c]), ap_base=int(ap_base[c]), ap_delta=ap_delta)
            for k in range(30):
                got = arena[int(off[c, k]):int(off[c, k]) + int(ln[c, k])].tobytes()
                assert got == want[k], (c, k, len(got), len(want[k]))
        # the streams do not overlap and follow each other in (container, content id) order
        flat_off, flat_len = off.reshape(-1), ln.reshape(-1)
        assert (flat_off[1:] >= flat_off[:-1] + flat_len[:-1]).all()
        # capacity guard: nothing is written and the device reports the reference-free code
        small = records.Writer(eng, DEV, n_cont, 64)
        small.arena.fill_(0xEE)
        small.write(reads, rc, feats)
        with pytest.raises(RuntimeError):
            small.lengths()
        assert np.array_equal(small.len.cpu().numpy().reshape(n_cont, 30), ln)
        assert bool((small.arena == 0xEE).all())
    finally:
        eng.close()


def test_writer_feeds_the_encoder(orc):
    """K3 -> F3 -> writer -> K1 on one container of synthetic reads: the writer's QS / BA / FC / FP / BS streams are what
    the oracle's writer makes of the oracle's features, and the rANS encoder takes them from the writer's arena."""
    import torch
    from cramtools_amd import external_compression as xc, records, synth, _lib
    rng = np.random.default_rng(7)
    ref = synth.reference(40000)
    rs = synth.make_reads(3000, ref, read_len=100, sub_rate=0.01, indel_rate=0.004, softclip_frac=0.05)
    pad = np.zeros(16, np.uint8)

    class R:
        pass
    o_reads = R()
    o_reads.n = rs.n
    for k in ("base_off", "read_len", "aln_start", "rflags", "cigar_off", "n_cigar"):
        setattr(o_reads, k, np.ascontiguousarray(getattr(rs, k)))
    o_reads.bases, o_reads.quals = np.concatenate([rs.bases, pad]), np.concatenate([rs.quals, pad])
    o_reads.cigar = np.concatenate([rs.cigar, np.zeros(1, np.uint32)])
    eng = xc.Engine(0)
    try:
        reads = records.ReadBatch(DEV, o_reads.bases, rs.base_off, rs.read_len, rs.aln_start, o_reads.cigar.view(np.int32),
                                  rs.cigar_off, rs.n_cigar, quals=o_reads.quals, mapq=rs.mapq, windows=((0, ref.size, 0),))
        d_ref = torch.as_tensor(np.concatenate([ref, np.zeros(16, np.uint8)])).to(DEV)
        fac = records.Sam2CramRecordFactory(eng)
        fac.set_ref_bases(d_ref)
        feats, hist = fac.create_variations(reads, feature_cap=3000 * 30)
        m = records.substitution_matrix(hist[0])
        fac.assign_codes(reads, feats, m)
        want_f = orc.extract_features(o_reads, ref)
        orc.assign_codes(want_f, orc.substitution_matrix(want_f["hist"]))
        n = reads.n
        cols = dict(bam_flags=rng.choice([99, 147, 83, 163], n), comp_flags=np.full(n, 3), read_group=np.full(n, -1),
                    mate_ref=np.zeros(n), mate_start=rng.integers(1, 40000, n), tlen=rng.integers(-500, 500, n),
                    tag_list=np.zeros(n))
        rc = records.RecordColumns(DEV, n, **cols)
        w = records.Writer(eng, DEV, 1, 4 * int(reads.bases.numel()) + 70 * n + 4096)
        w.write(reads, rc, feats)
        off, ln = w.lengths()
        full = {k: np.asarray(cols.get(k, np.zeros(n)), np.int32) for k in records.RECORD_COLS}
        want = orc.write_series(full, o_reads.read_len, o_reads.aln_start, rs.mapq, o_reads.bases, o_reads.quals,
                                o_reads.base_off, want_f)
        for name in ("QS", "BA", "FC", "FP", "BS", "FN", "AP", "TS", "NP", "IN", "SC", "DL", "MQ"):
            assert w.series(0, name) == want[_lib.SERIES_ID[name]], name
        # the rANS series straight from the writer's arena (descriptors from its two tables)
        ids = [(_lib.SERIES_ID[s], o) for s, o in (("QS", 1), ("AP", 0), ("BA", 1), ("BF", 1), ("CF", 1), ("TS", 1))]
        lens = [int(ln[0, k]) for k, _ in ids]
        blocks, _, out_bytes = xc.Engine.plan_encode(lens, [o for _, o in ids])
        for i, (k, _) in enumerate(ids):
            blocks[i].in_off = int(off[0, k])
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=DEV)
        d_len = torch.zeros(len(ids), dtype=torch.int32, device=DEV)
        d_pos = torch.zeros(len(ids), dtype=torch.int64, device=DEV)
        eng.encode_dev(w.arena, d_out, blocks, len(ids), d_len, d_pos)
        for i, (k, o) in enumerate(ids):
            got = d_out[int(d_pos[i]):int(d_pos[i]) + int(d_len[i])].cpu().numpy().tobytes()
            assert got == orc.rans(want[k], o), k
        # CRAMGPU_WS_ALIAS_QS: every record keeps its scores and the arrays lie back to back -> QS is not copied, its
        # offset points at reads.quals; everything else as before; the encoder reads both buffers from the lower address
        rc2 = records.RecordColumns(DEV, n, alias_qs=True, **cols)
        w2 = records.Writer(eng, DEV, 1, 4 * int(reads.bases.numel()) + 70 * n + 4096)
        w2.arena.fill_(0xEE)
        w2.write(reads, rc2, feats)
        off2, ln2 = w2.lengths()
        qs = _lib.SERIES_ID["QS"]
        assert np.array_equal(ln2, ln)
        assert w2.arena.data_ptr() + int(off2[0, qs]) == reads.quals.data_ptr() + int(rs.base_off[0])
        for name in ("BA", "FC", "FP", "BS", "FN", "AP", "TS", "NP", "IN", "SC", "DL", "MQ"):
            assert w2.series(0, name) == want[_lib.SERIES_ID[name]], name
        used = int(sum((int(ln2[0, k]) + 15) // 16 * 16 for k in range(30) if k != qs))
        assert bool((w2.arena[used:] == 0xEE).all())  # no slot for QS, nothing written behind the other streams
        base = min(w2.arena.data_ptr(), reads.quals.data_ptr()) & ~15
        for i, (k, _) in enumerate(ids):
            blocks[i].in_off = int(off2[0, k]) + w2.arena.data_ptr() - base
        d_out.zero_()
        eng.encode_dev(base, d_out, blocks, len(ids), d_len, d_pos)
        for i, (k, o) in enumerate(ids):
            got = d_out[int(d_pos[i]):int(d_pos[i]) + int(d_len[i])].cpu().numpy().tobytes()
            assert got == orc.rans(want[k], o), k
        # cramgpu_rans_encode_series_dev: the same blocks, their table made on the device behind the writer (both writers:
        # in_shift rebases the aliased QS stream), slots of cramgpu_rans_bound bytes one after the other
        for wr, shift_base in ((w, w.arena.data_ptr()), (w2, base)):
            d_blk = torch.zeros(32 * len(ids), dtype=torch.uint8, device=DEV)
            d_st = torch.full((1,), 77, dtype=torch.int32, device=DEV)
            d_out.zero_()
            eng.encode_series_dev(shift_base, wr.arena.data_ptr() - shift_base, wr.off, wr.len, wr.status, 1,
                                  [k for k, _ in ids], [o for _, o in ids], d_out, d_blk, d_len, d_pos, d_st)
            assert int(d_st.item()) == 0
            tab = d_blk.cpu().numpy().view(_lib.BLOCK_DTYPE)
            o_w, l_w = wr.lengths()
            at = 0
            for i, (k, o) in enumerate(ids):
                assert (int(tab["in_off"][i]), int(tab["in_len"][i]), int(tab["order"][i])) == \
                    (int(o_w[0, k]) + wr.arena.data_ptr() - shift_base, int(l_w[0, k]), o)
                assert int(tab["out_cap"][i]) == _lib.lib().cramgpu_rans_bound(int(l_w[0, k])) and int(tab["out_off"][i]) == at
                at += (int(tab["out_cap"][i]) + 15) // 16 * 16
                got = d_out[int(d_pos[i]):int(d_pos[i]) + int(d_len[i])].cpu().numpy().tobytes()
                assert got == orc.rans(want[k], o), k
        # slots that do not fit the output arena: a status, and every block left empty
        d_st.fill_(77)
        d_len.fill_(5)
        eng.encode_series_dev(w.arena.data_ptr(), 0, w.off, w.len, w.status, 1, [k for k, _ in ids], [o for _, o in ids],
                              d_out[:at - 16], d_blk, d_len, d_pos, d_st)
        assert int(d_st.item()) == _lib.E_INSUFFICIENT and bool((d_len == 0).all())
    finally:
        eng.close()


def test_encoder_behind_the_writer_many_blocks(orc):
    """cramgpu_rans_encode_series_dev over more blocks than the 4096 an LDS sort holds (the longest-first order is a radix
    sort on the device): 600 containers x 8 series of random short streams, every stream against the oracle."""
    import torch
    from cramtools_amd import external_compression as xc, _lib
    rng = np.random.default_rng(11)
    n_cont, ns = 600, _lib.N_SERIES
    ids = [4, 1, 6, 0, 21, 14, 9, 25]
    orders = [1, 0, 1, 1, 1, 1, 1, 0]
    ln = np.zeros((n_cont, ns), np.int32)
    ln[:, ids] = rng.integers(0, 400, (n_cont, len(ids)))
    ln[rng.integers(0, n_cont, 40), rng.choice(ids, 40)] = 0        # empty streams among them
    ln[5, 4], ln[77, 6] = 70000, 30000                               # and two long ones
    slot = (ln.astype(np.int64) + 15) // 16 * 16
    off = (np.cumsum(slot.reshape(-1)) - slot.reshape(-1)).reshape(n_cont, ns)
    arena = rng.integers(0, 7, int(slot.sum()) + 16).astype(np.uint8) * 9
    eng = xc.Engine(0)
    try:
        d_arena = torch.as_tensor(arena).to(DEV)
        d_off, d_ln = torch.as_tensor(off.reshape(-1)).to(DEV), torch.as_tensor(ln.reshape(-1)).to(DEV)
        n = n_cont * len(ids)
        caps = np.array([_lib.lib().cramgpu_rans_bound(int(x)) for x in ln[:, ids].reshape(-1)], np.int64)
        d_out = torch.zeros(int(((caps + 15) // 16 * 16).sum()) + 16, dtype=torch.uint8, device=DEV)
        d_blk = torch.zeros(32 * n, dtype=torch.uint8, device=DEV)
        d_len, d_pos = torch.zeros(n, dtype=torch.int32, device=DEV), torch.zeros(n, dtype=torch.int64, device=DEV)
        d_st = torch.full((1,), 77, dtype=torch.int32, device=DEV)
        for _ in range(2):  # twice: the workspaces of the first call are reused
            eng.encode_series_dev(d_arena, 0, d_off, d_ln, None, n_cont, ids, orders, d_out, d_blk, d_len, d_pos, d_st)
        assert int(d_st.item()) == 0
        lens, pos, out = d_len.cpu().numpy(), d_pos.cpu().numpy(), d_out.cpu().numpy()
        for c in range(n_cont):
            for k, (sid, o) in enumerate(zip(ids, orders)):
                b = c * len(ids) + k
                raw = arena[off[c, sid]:off[c, sid] + ln[c, sid]].tobytes()
                assert out[pos[b]:pos[b] + lens[b]].tobytes() == orc.rans(raw, o), (c, sid)
    finally:
        eng.close()


@pytest.mark.parametrize("seed,sizes", [(0, [1]), (1, [300]), (2, [700, 5, 256, 257]), (3, [64, 1, 1000])])
def test_reader_matches_oracle_and_inverts_writer(orc, seed, sizes):
    """writer -> reader on the device: every column, feature and array equals what the oracle's CramRecordReader#read makes
    of the oracle's streams, container by container."""
    import torch
    from cramtools_amd import external_compression as xc, records
    rng = np.random.default_rng(200 + seed)
    reads, cols, feats, parts = _batch(rng, sizes)
    n_cont, n = len(sizes), reads.n
    multiref = np.array([(c + seed) & 1 for c in range(n_cont)], np.uint8)
    ap_base = np.array([int(p["aln_start"][0]) - (c % 3) for c, p in enumerate(parts)], np.int32)
    ap_delta = seed != 3
    eng = xc.Engine(0)
    try:
        rc = records.RecordColumns(DEV, n, multiref=multiref, ap_base=ap_base, ap_delta=ap_delta, **cols)
        w = records.Writer(eng, DEV, n_cont, 8 * int(reads.bases.numel()) + 70 * n + 4096)
        w.write(reads, rc, feats)
        w.lengths()
        cont_first = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(DEV)
        rd = records.CramRecordReader(eng, DEV, n, n_cont, int(reads.bases.numel()), feats.count + 8, int(reads.bases.numel()))
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
        rd.check()
        got_f = rd.feats.to_host()
        r0 = f0 = 0
        base_off, seq_off = rd.base_off.cpu().numpy(), rd.seq_off.cpu().numpy()
        g_bases, g_quals, g_seq = rd.bases.cpu().numpy(), rd.quals.cpu().numpy(), rd.seq.cpu().numpy()
        for c, p in enumerate(parts):
            streams = orc.write_series(p["cols"], p["read_len"], p["aln_start"], p["mapq"], p["bases"], p["quals"], p["base_off"],
                                       p["feats"], multiref=bool(multiref[c]), ap_base=int(ap_base[c]), ap_delta=ap_delta)
            want = orc.read_series(streams, p["n"], multiref=bool(multiref[c]), ap_base=int(ap_base[c]), ap_delta=ap_delta)
            m, nf = p["n"], want["n_feat"]
            for k in records.RECORD_COLS:
                assert np.array_equal(getattr(rd.cols, k)[r0:r0 + m].cpu().numpy(), want[k]), (c, k)
            assert np.array_equal(rd.read_len[r0:r0 + m].cpu().numpy(), want["read_len"])
            assert np.array_equal(rd.aln_start[r0:r0 + m].cpu().numpy(), want["aln_start"])
            assert np.array_equal(rd.mapq[r0:r0 + m].cpu().numpy(), want["mapq"])
            assert np.array_equal(base_off[r0:r0 + m + 1] - base_off[r0], want["base_off"])
            assert np.array_equal(seq_off[r0:r0 + m + 1] - seq_off[r0], want["seq_off"])
            assert np.array_equal(got_f["off"][r0:r0 + m + 1] - f0, want["feat_off"])
            for k in ("op", "pos", "a", "b"):
                assert np.array_equal(got_f[k][f0:f0 + nf], want[k][:nf]), (c, k)
            nb, nsq = int(want["base_off"][m]), int(want["seq_off"][m])
            b0, s0 = int(base_off[r0]), int(seq_off[r0])
            assert np.array_equal(g_quals[b0:b0 + nb], want["quals"][:nb])
            assert np.array_equal(g_seq[s0:s0 + nsq], want["seq"][:nsq])
            unm = (want["bam_flags"] & 4) != 0
            for r in np.nonzero(unm & ((want["comp_flags"] & 8) == 0))[0]:
                o, L = int(want["base_off"][r]), int(want["read_len"][r])
                assert np.array_equal(g_bases[b0 + o:b0 + o + L], want["bases"][o:o + L])
            r0 += m
            f0 += nf
        assert rd.feats.count == f0
        # a stream cut short, a capacity too small: reported, no fault
        w.len[3] -= 1  # container 0, FC... any series: the reader must notice
        w.len[0] -= 1
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
        with pytest.raises(RuntimeError):
            rd.check()
        w.len[3] += 1
        w.len[0] += 1
        # the value workspace is sized from the caller's byte count, the parse from the device
        # lengths: lengths that sum to more than that, or a negative one, are reported before anything is parsed
        if n > 50:  # (a one-record container fits the slack every workspace has)
            rd.read(w.arena, w.off, w.len, cont_first, 16, multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
            with pytest.raises(RuntimeError):
                rd.check()
        keep = int(w.len[5])
        w.len[5] = -8
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
        with pytest.raises(RuntimeError):
            rd.check()
        w.len[5] = keep
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
        rd.check()  # and the reader works as before afterwards
        if feats.count > 4:
            small = records.CramRecordReader(eng, DEV, n, n_cont, int(reads.bases.numel()), 2, int(reads.bases.numel()))
            small.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
            with pytest.raises(RuntimeError):
                small.check()
    finally:
        eng.close()


@pytest.mark.parametrize("seed,sizes,capture", [(0, [1], True), (1, [300, 7], False), (2, [700, 5, 256, 257], True), (3, [64, 1, 1000], False)])
def test_read_names_series(orc, seed, sizes, capture):
    """RN: the writer puts a record's name + '\\t' for every record (names captured) or for the detached ones (Writer#write;
    ReaderToBAM.java:86-93), the reader hands every record the place of its name in the stream; both against the oracle.
    Names of 0 to 40 bytes lie in an arena of their own in shuffled order (offsets, as a BAM stripe gives them)."""
    import torch
    from cramtools_amd import external_compression as xc, records
    rng = np.random.default_rng(300 + seed)
    reads, cols, feats, parts = _batch(rng, sizes)
    n_cont, n = len(sizes), reads.n
    names = [bytes(rng.integers(33, 127, int(rng.integers(0, 41)), dtype=np.uint8)) for _ in range(n)]
    order = rng.permutation(n)
    name_off, at = np.zeros(n, np.int64), 3
    blob = bytearray(b"xyz")
    for r in order:
        name_off[r] = at
        blob += names[r] + b"#"
        at += len(names[r]) + 1
    name_len = np.array([len(x) for x in names], np.int32)
    d_names = torch.as_tensor(np.frombuffer(bytes(blob) + bytes(16), np.uint8).copy()).to(DEV)
    d_noff, d_nlen = torch.as_tensor(name_off).to(DEV), torch.as_tensor(name_len).to(DEV)
    eng = xc.Engine(0)
    try:
        rc = records.RecordColumns(DEV, n, names=d_names, name_off=d_noff, name_len=d_nlen, capture_names=capture, **cols)
        w = records.Writer(eng, DEV, n_cont, 8 * int(reads.bases.numel()) + 120 * n + 4096)
        w.write(reads, rc, feats)
        off, ln = w.lengths()
        arena = w.arena.cpu().numpy()
        r0 = 0
        all_names = []
        for c, p in enumerate(parts):
            want = orc.write_series(p["cols"], p["read_len"], p["aln_start"], p["mapq"], p["bases"], p["quals"], p["base_off"],
                                    p["feats"], names=names[r0:r0 + p["n"]], capture_names=capture)
            for k in range(30):
                got = arena[int(off[c, k]):int(off[c, k]) + int(ln[c, k])].tobytes()
                assert got == want[k], (c, k, len(got), len(want[k]))
            all_names += orc.read_names(want, p["cols"]["comp_flags"], capture)
            r0 += p["n"]
        cont_first = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(DEV)
        rd = records.CramRecordReader(eng, DEV, n, n_cont, int(reads.bases.numel()), feats.count + 8, int(reads.bases.numel()))
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), capture_names=capture)
        rd.check()
        g_off, g_len = rd.name_off.cpu().numpy(), rd.name_len.cpu().numpy()
        for r in range(n):
            if all_names[r] is None:
                assert g_len[r] == -1, r
            else:
                assert arena[int(g_off[r]):int(g_off[r]) + int(g_len[r])].tobytes() == all_names[r] == names[r], r
        # a container written WITHOUT names reads back with no names and no complaint
        rc2 = records.RecordColumns(DEV, n, **cols)
        w.write(reads, rc2, feats)
        w.lengths()
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()))
        rd.check()
        assert bool((rd.name_len == -1).all().item())
        # one name short: the reader notices
        rc3 = records.RecordColumns(DEV, n, names=d_names, name_off=d_noff, name_len=d_nlen, capture_names=True, **cols)
        w.write(reads, rc3, feats)
        w.lengths()
        last = int(ln.shape[0]) - 1
        cut = len(names[n - 1]) + 1
        w.len[last * 30 + 12] -= cut
        rd.read(w.arena, w.off, w.len, cont_first, int(w.arena.numel()), capture_names=True)
        with pytest.raises(RuntimeError):
            rd.check()
    finally:
        eng.close()
