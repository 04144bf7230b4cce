"""Container assembly and BAM records on the device (cramgpu_cigars_dev, cramgpu_restore_mates_dev,
cramgpu_bam_encode_dev, cramgpu_bam_decode_dev) against the oracle (oracle/bam.c): random
containers with every operator and record kind, shared and out-of-range mate links, names / tags present and generated,
records without scores, bases outside the 4-bit table, the capacity guards, and encode -> decode round trips."""
import numpy as np
import pytest

from tests import bam_cases as bc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(x, dt=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(x))
    return (t.to(dt) if dt is not None else t).to(DEV)


def _setup(b):
    from cramtools_amd import records
    import torch
    f = b["feats"]
    feats = records.FeatureBatch.from_host(DEV, f["op"], f["pos"], f["a"], f["b"], f["off"])
    cols = records.RecordColumns(DEV, b["n"], multiref=b["multiref"], **b["cols"])
    return feats, cols, _t(b["read_len"]), _t(b["aln_start"]), torch


@pytest.mark.parametrize("seed,sizes", [(0, [1]), (1, [300]), (2, [700, 5, 256, 257]), (3, [64, 1, 1000])])
def test_assembly_and_encode_match_oracle(orc, seed, sizes):
    from cramtools_amd import bam, external_compression as xc
    rng = np.random.default_rng(200 + seed)
    b = bc.random_batch(rng, sizes)
    c = b["cols"]
    eng = xc.Engine(0)
    try:
        feats, cols, read_len, aln_start, torch = _setup(b)
        fac = bam.Cram2SamRecordFactory(eng, DEV)
        # --- CIGARs and alignment ends
        want = orc.cigars(b["read_len"], b["aln_start"], c["bam_flags"], b["feats"])
        n_cigar, cigar_off, cigar, aln_end = fac.cigars(read_len, aln_start, cols.bam_flags, feats)
        assert np.array_equal(n_cigar.cpu().numpy(), want["n_cigar"])
        assert np.array_equal(cigar_off.cpu().numpy(), want["cigar_off"])
        assert np.array_equal(cigar.cpu().numpy().view(np.uint32)[:want["total"]], want["cigar"][:want["total"]])
        assert np.array_equal(aln_end.cpu().numpy(), want["aln_end"])
        if want["total"] > 1:
            with pytest.raises(bam._lib.CramGpuError) as ei:
                fac.cigars(read_len, aln_start, cols.bam_flags, feats, cigar_cap=want["total"] - 1)
            assert ei.value.code == -3
        # --- mates
        w_cols, w_st = orc.restore_mates(b["cont_first"], c, b["ref_id"], b["aln_start"], want["aln_end"])
        st = fac.restore_mates(cols, _t(b["cont_first"]), _t(b["cont_ref_id"]), aln_start, aln_end)
        assert np.array_equal(st.cpu().numpy(), w_st)
        assert np.array_equal(cols.ref_id.cpu().numpy(), b["ref_id"])
        good = (w_st == 0)[b["cont"]]  # a container the reference throws on: next_frag is unspecified, the mate columns stay
        for k in ("next_frag", "mate_flags", "mate_ref", "mate_start", "tlen"):
            sel = good if k == "next_frag" else slice(None)
            assert np.array_equal(getattr(cols, k).cpu().numpy()[sel], w_cols[k][sel]), k
        # --- BAM records: names and tags from the host, a fifth of the records without scores
        codec = bam.BAMRecordCodec(eng, DEV)
        qual_status = (rng.random(b["n"]) < 0.2).astype(np.int32)
        if seed == 2:
            b["bases"] = b["bases"].copy()
            b["bases"][rng.integers(0, int(b["base_off"][-1]), 30)] = rng.choice(np.frombuffer(b"=acgtnMRSVWYHKDB.@x", np.uint8), 30)
        args = (b["ref_id"], b["aln_start"], want["aln_end"], b["mapq"], c["bam_flags"], w_cols["mate_flags"], w_cols["mate_ref"],
                w_cols["mate_start"], w_cols["tlen"], b["read_len"], b["base_off"], b["bases"], b["quals"], 1 - qual_status,
                want["n_cigar"], want["cigar_off"], want["cigar"])
        w_bam, w_off, w_status, tot = orc.bam_encode(*args, names=b["names"], name_off=b["name_off"], tags=b["tags"], tag_off=b["tag_off"])
        bases, quals, base_off, mapq = _t(b["bases"]), _t(b["quals"]), _t(b["base_off"]), _t(b["mapq"])
        enc = lambda cap, **kw: codec.encode(cols, read_len, aln_start, aln_end, mapq, base_off, bases, quals, n_cigar, cigar_off,
                                             cigar, cap, qual_status=_t(qual_status), **kw)
        kw = dict(names=_t(b["names"]), name_off=_t(b["name_off"]), tags=_t(b["tags"]), tag_off=_t(b["tag_off"]))
        g_bam, g_off, g_status = enc(tot, **kw)
        assert np.array_equal(g_off.cpu().numpy(), w_off)
        assert np.array_equal(g_status.cpu().numpy(), w_status)
        assert (w_status != 0).any() == (seed == 2)
        assert g_bam[:tot].cpu().numpy().tobytes() == w_bam
        # capacity guard: nothing written, every record says so, the need is reported
        g_bam, g_off, g_status = enc(tot - 1, **kw)
        assert int(g_off[-1].item()) == tot and (g_status.cpu().numpy() == -3).all() and not g_bam.any().item()
        # generated names, no tags
        w_bam2, w_off2, _, tot2 = orc.bam_encode(*args, name_base=10 ** 9 - 5)
        g_bam, g_off, _ = enc(tot2, name_base=10 ** 9 - 5)
        assert np.array_equal(g_off.cpu().numpy(), w_off2) and g_bam[:tot2].cpu().numpy().tobytes() == w_bam2
        # --- and back: the stripe decodes to what the oracle's decoder says
        stripe = np.frombuffer(w_bam, np.uint8)
        rec_off = codec.index(stripe)
        assert np.array_equal(rec_off, w_off)
        wd = orc.bam_decode(stripe, rec_off)
        nb, ncg = int(wd["base_off"][-1]), int(wd["cigar_off"][-1])
        d = codec.decode(_t(np.concatenate([stripe, np.zeros(16, np.uint8)])), _t(rec_off), nb, ncg)
        for k in ("ref_id", "aln_start", "mapq", "bam_flags", "mate_ref", "mate_start", "tlen", "read_len", "base_off", "n_cigar",
                  "cigar_off", "aln_span", "rflags", "name_off", "name_len", "tag_off", "tag_len", "status"):
            assert np.array_equal(d[k].cpu().numpy(), wd[k]), k
        assert np.array_equal(d["bases"][:nb].cpu().numpy(), wd["bases"][:nb])
        assert np.array_equal(d["quals"][:nb].cpu().numpy(), wd["quals"][:nb])
        assert np.array_equal(d["cigar"].cpu().numpy().view(np.uint32)[:ncg], wd["cigar"][:ncg])
        if nb > 0:
            d = codec.decode(_t(np.concatenate([stripe, np.zeros(16, np.uint8)])), _t(rec_off), nb - 1, ncg)
            assert (d["status"].cpu().numpy() == -3).all() and int(d["base_off"][-1].item()) == nb
    finally:
        eng.close()


def test_decoder_flags_malformed_records(orc):
    from cramtools_amd import bam, external_compression as xc
    rng = np.random.default_rng(5)
    recs = [bc.py_bam_record(1, 100 + i, 20, 0, 0, "n%d" % i, [(10 + i) << 4], "ACGTACGTAC" + "G" * i, bytes(range(10 + i)), -1, 0, 0)
            for i in range(6)]
    bad = bytearray(recs[2])
    bad[20:24] = (10 ** 6).to_bytes(4, "little")  # l_seq runs past the record
    recs[2] = bytes(bad)
    stripe = np.frombuffer(b"".join(recs), np.uint8)
    rec_off = orc.bam_index(stripe)
    wd = orc.bam_decode(stripe, rec_off)
    assert wd["status"].tolist() == [0, 0, -5, 0, 0, 0]
    eng = xc.Engine(0)
    try:
        codec = bam.BAMRecordCodec(eng, DEV)
        d = codec.decode(_t(np.concatenate([stripe, np.zeros(16, np.uint8)])), _t(rec_off), int(wd["base_off"][-1]), 6)
        for k in ("status", "read_len", "base_off", "rflags", "aln_start", "n_cigar"):
            assert np.array_equal(d[k].cpu().numpy(), wd[k]), k
        nb = int(wd["base_off"][-1])
        assert np.array_equal(d["bases"][:nb].cpu().numpy(), wd["bases"][:nb])
        with pytest.raises(bam._lib.CramGpuError):
            codec.index(stripe[:-3])
    finally:
        eng.close()


def test_slice_md5_matches_hashlib(orc):
    """RFC 1321's own vectors, every tail length around the padding boundaries, unaligned starts, spans clipped at the end of
    the window, a start beyond it, an unplaced slice."""
    from cramtools_amd import bam, external_compression as xc
    import torch
    rng = np.random.default_rng(3)
    ref = rng.choice(np.frombuffer(b"ACGTN", np.uint8), 70000)
    ref[:3] = np.frombuffer(b"abc", np.uint8)
    msg = b"12345678901234567890123456789012345678901234567890123456789012345678901234567890"
    ref[100:100 + len(msg)] = np.frombuffer(msg, np.uint8)
    cases = [(1, 3, 0, 70000, 0), (1, 0, 0, 70000, 0), (101, len(msg), 0, 70000, 0)]
    cases += [(7, L, 0, 70000, 0) for L in list(range(50, 70)) + [119, 120, 121, 127, 128, 129]]
    cases += [(int(rng.integers(1, 5000)), int(rng.integers(0, 66000)), 0, 70000, 0) for _ in range(40)]
    cases += [(69990, 500, 0, 70000, 0), (70001, 10, 0, 70000, 0), (0, 10, 0, 70000, 0), (-5, 0, 0, 70000, 0)]
    cases += [(1500, 900, 1000, 2000, 1200), (2500, 900, 1000, 2000, 1200), (3201, 5, 1000, 2000, 1200)]  # windows inside the arena
    want = [orc.slice_md5(ref[off:off + ln], st, sp, s0) for st, sp, off, ln, s0 in cases]
    assert want[0].hex() == "900150983cd24fb0d6963f7d28e17f72" and want[1].hex() == "d41d8cd98f00b204e9800998ecf8427e"
    assert want[2].hex() == "57edf4a22be3c955ac49da2e2107b67a"
    eng = xc.Engine(0)
    try:
        d_ref = _t(np.concatenate([ref, np.zeros(16, np.uint8)]))
        got = bam.slice_ref_md5(eng, DEV, d_ref, [(c[2], c[3], c[4]) for c in cases], [c[0] for c in cases], [c[1] for c in cases])
        got = got.cpu().numpy()
        for i, w in enumerate(want):
            assert got[i].tobytes() == w, cases[i]
    finally:
        eng.close()



def test_mates_and_cigars_at_scale(orc):
    """500 000 records in 50 containers, pairs linked by next-fragment distances (mates 0..40 records downstream, some
    three-fragment chains, some detached), features on a third of the reads: mate columns, CIGARs and alignment ends equal
    the oracle's for every record."""
    from cramtools_amd import bam, external_compression as xc, records
    rng = np.random.default_rng(77)
    n_cont, per = 50, 10000
    n = n_cont * per
    bf = np.zeros(n, np.int32)
    cf = np.zeros(n, np.int32)
    nf = np.zeros(n, np.int32)
    taken = np.zeros(n, bool)
    for c in range(n_cont):
        lo = c * per
        i = 0
        while i < per:
            r = lo + i
            roll = rng.random()
            if taken[r] or roll < 0.1:
                i += 1
                continue
            d = int(rng.integers(0, 41))
            j = r + 1 + d
            if roll < 0.2:  # detached pair member
                bf[r], cf[r] = 1 | 64, 2
            elif j < lo + per and not taken[j]:
                bf[r] = 1 | 64 | (16 if rng.random() < 0.5 else 0)
                cf[r], nf[r] = 4, d
                bf[j] = 1 | 128 | (16 if rng.random() < 0.5 else 0) | (4 if rng.random() < 0.02 else 0)
                taken[j] = True
                if rng.random() < 0.05 and j + 3 < lo + per and not taken[j + 3]:  # a third fragment
                    cf[j], nf[j] = 4, 2
                    bf[j + 3] = 1 | (16 if rng.random() < 0.5 else 0)
                    taken[j + 3] = True
            taken[r] = True
            i += 1
    read_len = np.full(n, 100, np.int32)
    aln_start = (np.sort(rng.integers(1, 1 << 24, n)) + 1).astype(np.int32)
    k = rng.random(n) < 0.33
    k &= (bf & 4) == 0
    nfeat = np.where(k, rng.integers(1, 4, n), 0)
    off = np.concatenate([[0], np.cumsum(nfeat)]).astype(np.int64)
    tot = int(off[-1])
    ops = rng.choice(np.frombuffer(b"XIDSNiQ", np.uint8), tot)
    rid = np.repeat(np.arange(n), nfeat)
    pos = (rng.integers(1, 90, tot) + 0).astype(np.int32)
    order = np.lexsort((pos, rid))
    pos, ops = pos[order], ops[order]
    fa = rng.integers(1, 30, tot).astype(np.int32)
    fb = rng.integers(1, 10, tot).astype(np.int32)
    feats_h = dict(off=off, op=ops, pos=pos, a=fa, b=fb)
    cols_h = {c_: np.zeros(n, np.int32) for c_ in bc.sc_cols()}
    cols_h.update(bam_flags=bf, comp_flags=cf, next_frag=nf, mate_flags=rng.integers(0, 4, n).astype(np.int32),
                  mate_ref=np.full(n, 9, np.int32), mate_start=np.full(n, 7, np.int32), tlen=np.full(n, 5, np.int32))
    cont_first = (np.arange(n_cont + 1) * per).astype(np.int32)
    cont_ref = rng.integers(0, 24, n_cont).astype(np.int32)
    ref_id = np.repeat(cont_ref, per)
    want_c = orc.cigars(read_len, aln_start, bf, feats_h)
    want_m, want_st = orc.restore_mates(cont_first, cols_h, ref_id, aln_start, want_c["aln_end"])
    assert not want_st.any()
    eng = xc.Engine(0)
    try:
        feats = records.FeatureBatch.from_host(DEV, ops, pos, fa, fb, off)
        cols = records.RecordColumns(DEV, n, **cols_h)
        fac = bam.Cram2SamRecordFactory(eng, DEV)
        n_cigar, cigar_off, cigar, aln_end = fac.cigars(_t(read_len), _t(aln_start), cols.bam_flags, feats)
        assert np.array_equal(n_cigar.cpu().numpy(), want_c["n_cigar"]) and np.array_equal(aln_end.cpu().numpy(), want_c["aln_end"])
        assert np.array_equal(cigar.cpu().numpy().view(np.uint32)[:want_c["total"]], want_c["cigar"][:want_c["total"]])
        st = fac.restore_mates(cols, _t(cont_first), _t(cont_ref), _t(aln_start), aln_end)
        assert not st.cpu().numpy().any()
        for k_ in ("next_frag", "mate_flags", "mate_ref", "mate_start", "tlen"):
            assert np.array_equal(getattr(cols, k_).cpu().numpy(), want_m[k_]), k_
        assert (want_m["tlen"] != 5).sum() > n // 4  # the chains really were restored
    finally:
        eng.close()


@pytest.mark.parametrize("seed,sizes", [(0, [1]), (1, [2, 3]), (2, [700, 5, 256, 257]), (3, [10000, 10000, 4000])])
def test_set_mate_info_matches_oracle(orc, seed, sizes):
    """cramgpu_set_mate_info_dev against the sequential restatement of CramSerilization.java:274-373: names scattered in a
    buffer (offset + length, as cramgpu_bam_decode_dev reports them), every template kind of bam_cases.mate_batch."""
    from cramtools_amd import bam, records, external_compression as xc
    import torch
    rng = np.random.default_rng(300 + seed)
    b = bc.mate_batch(rng, sizes)
    c = b["cols"]
    want_cf, want_nf = orc.set_mate_info(b["cont_first"], c, c["ref_id"], b["aln_start"], b["aln_end"], b["names"], b["name_off"])
    # the names as a BAM stripe holds them: not back to back, in another order
    n = b["n"]
    lens = np.diff(b["name_off"]).astype(np.int32)
    perm = rng.permutation(n)
    gaps = rng.integers(1, 40, n)
    pos = np.zeros(n, np.int64)
    at = 0
    for r in perm:
        at += int(gaps[r])
        pos[r] = at
        at += int(lens[r])
    buf = rng.integers(0, 256, at + 8).astype(np.uint8)
    for r in range(n):
        buf[pos[r]:pos[r] + lens[r]] = b["names"][b["name_off"][r]:b["name_off"][r + 1]]
    eng = xc.Engine(0)
    try:
        cols = records.RecordColumns(DEV, n, **c)
        ser = bam.CramSerilization(eng, DEV)
        ser.set_mate_info(cols, _t(b["cont_first"]), _t(b["aln_start"]), _t(b["aln_end"]), _t(buf), _t(pos), _t(lens))
        torch.cuda.synchronize()
        assert np.array_equal(cols.comp_flags.cpu().numpy(), want_cf)
        assert np.array_equal(cols.next_frag.cpu().numpy(), want_nf)
        for k in ("bam_flags", "mate_flags", "mate_ref", "mate_start", "tlen", "ref_id"):
            assert np.array_equal(getattr(cols, k).cpu().numpy(), c[k]), k
        # a second call on its own output changes nothing (the links are a function of the names and the mate fields)
        ser.set_mate_info(cols, _t(b["cont_first"]), _t(b["aln_start"]), _t(b["aln_end"]), _t(buf), _t(pos), _t(lens))
        torch.cuda.synchronize()
        assert np.array_equal(cols.comp_flags.cpu().numpy(), want_cf)
    finally:
        eng.close()


@pytest.mark.parametrize("seed,sizes,max_len", [(0, [1, 0, 3], 60), (1, [700, 256], 300), (2, [40], 20000)])
def test_content_digests_match_oracle(orc, seed, sizes, max_len):
    from cramtools_amd import bam, external_compression as xc
    import torch
    rng = np.random.default_rng(400 + seed)
    n = sum(sizes)
    read_len = rng.integers(0, max_len + 1, n).astype(np.int32)
    base_off = np.concatenate([[0], np.cumsum(read_len)]).astype(np.int64)
    bases = rng.integers(0, 256, int(base_off[-1]) + 16).astype(np.uint8)
    quals = rng.integers(0, 94, int(base_off[-1]) + 16).astype(np.uint8)
    cont_first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    want = orc.content_digests(cont_first, base_off, read_len, bases, quals)
    eng = xc.Engine(0)
    try:
        ser = bam.CramSerilization(eng, DEV)
        got = ser.content_digests(_t(cont_first), _t(base_off), _t(read_len), _t(bases), _t(quals))
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().astype(np.uint32), want)
        tags = bam.slice_content_tags(int(got[0, 0]), int(got[0, 1]))
        import hashlib
        assert [t for t, _ in tags] == ["BD", "SD", "B5", "S5", "B1", "S1"]
        assert tags[0][1] == int(want[0, 0]).to_bytes(4, "little") and tags[2][1] == hashlib.sha512(b"").digest()
        assert tags[5][1] == hashlib.sha1(b"").digest()
    finally:
        eng.close()


def test_reference_bam_record_test_inputs():
    """The records the reference's own tests build (TestBAMRecordView.java:47-66, BAMRecordViewTest.java:78-124) through the
    device codec; expected bytes from the independent statement of BAMRecordView's layout in tests/bam_cases.py."""
    import torch
    from cramtools_amd import bam, external_compression as xc, records
    eng = xc.Engine(0)
    try:
        codec = bam.BAMRecordCodec(eng, DEV)
        r = bc.REF_VIEW_RECORD
        i32 = lambda *v: np.array(v, np.int32)
        cols = records.RecordColumns(DEV, 1, bam_flags=i32(555 & ~0x28), ref_id=i32(0), mate_flags=i32(3), mate_ref=i32(0),
                                     mate_start=i32(78), tlen=i32(133))
        pad = lambda b: _t(np.frombuffer(b + bytes(16), np.uint8))
        want = bc.ref_view_record_bytes(bc.reg2bin(76, 86))
        g_bam, g_off, g_st = codec.encode(cols, _t(i32(10)), _t(i32(77)), _t(i32(86)), _t(np.array([44], np.uint8)),
                                          _t(np.array([0, 10], np.int64)), pad(r["seq"].encode()), pad(r["qual"]), _t(i32(1)),
                                          _t(np.array([0, 1], np.int64)), _t(np.array(r["cigar"], np.int32)), len(want),
                                          names=pad(b"name1"), name_off=_t(np.array([0, 5], np.int64)), tags=pad(r["tags"]),
                                          tag_off=_t(np.array([0, 4], np.int64)))
        assert g_bam[:len(want)].cpu().numpy().tobytes() == want and int(g_st[0].item()) == 0 and g_off.tolist() == [0, len(want)]
        raw = bc.ref_view_record_bytes(99)  # as the reference's test leaves it: index bin 99, set by hand
        d = codec.decode(pad(raw), _t(np.array([0, len(raw)], np.int64)), 10, 1)
        one = lambda k: int(d[k][0].item())
        assert (one("ref_id"), one("aln_start"), one("mapq"), one("bam_flags")) == (0, 77, 44, 555) and one("status") == 0
        assert (one("mate_ref"), one("mate_start"), one("tlen"), one("read_len")) == (0, 78, 133, 10)
        assert d["bases"][:10].cpu().numpy().tobytes() == b"A" * 10 and d["quals"][:10].cpu().numpy().tobytes() == b"B" * 10
        assert raw[one("name_off"):one("name_off") + one("name_len")] == b"name1"
        assert raw[one("tag_off"):one("tag_off") + one("tag_len")] == b"AMAQ"
        # the unmapped record, twice ("readName", "2"), in one stripe
        cols = records.RecordColumns(DEV, 2, bam_flags=i32(4, 4), ref_id=i32(-1, -1), mate_ref=i32(-1, -1))
        want = bc.ref_unmapped_pair_bytes()
        g_bam, g_off, g_st = codec.encode(cols, _t(i32(1, 1)), _t(i32(0, 0)), _t(i32(0, 0)), _t(np.zeros(2, np.uint8)),
                                          _t(np.array([0, 1, 2], np.int64)), pad(b"AA"), pad(b"\0\0"), _t(i32(0, 0)),
                                          _t(np.array([0, 0, 0], np.int64)), _t(np.zeros(1, np.int32)), len(want),
                                          names=pad(b"readName2"), name_off=_t(np.array([0, 8, 9], np.int64)))
        assert g_bam[:len(want)].cpu().numpy().tobytes() == want and not g_st.any().item()
        rec_off = codec.index(np.frombuffer(want, np.uint8))
        assert rec_off.tolist() == [0, 47, len(want)]
        d = codec.decode(pad(want), _t(rec_off), 2, 0)
        assert d["ref_id"].tolist() == [-1, -1] and d["bam_flags"].tolist() == [4, 4] and d["read_len"].tolist() == [1, 1]
        assert d["status"].tolist() == [0, 0] and d["quals"][:2].tolist() == [0, 0]
    finally:
        eng.close()
        torch.cuda.empty_cache()


def _index_both(codec, orc, stripe, n_ref=64, cap=None):
    want = orc.bam_index(stripe)
    cap = len(want) if cap is None else cap
    d = _t(np.concatenate([np.frombuffer(bytes(stripe), np.uint8), np.zeros(16, np.uint8)]))
    rec_off, n, st = codec.index_dev(d, len(stripe), n_ref, cap)
    assert st == 0 and n == len(want) - 1, (st, n, len(want) - 1)
    k = min(cap, len(want))
    assert np.array_equal(rec_off[:k].cpu().numpy(), want[:k])
    return want


@pytest.mark.parametrize("seed,sizes", [(0, [1]), (1, [300]), (2, [700, 5, 256, 257]), (3, [64, 1, 3000])])
def test_record_boundaries_on_the_device(orc, seed, sizes):
    """cramgpu_bam_index_dev against the sequential walk (oracle / Bam_OBA_Supplier.java:49-62): random stripes over many
    4 KiB segments; records much longer than a segment; bytes that read like record heads inside tags, right where a segment
    starts (the guess is wrong there and the verification pass must repair it); a capacity too small; a broken block size."""
    from cramtools_amd import bam, external_compression as xc
    rng = np.random.default_rng(500 + seed)
    b = bc.random_batch(rng, sizes)
    c = b["cols"]
    cg = orc.cigars(b["read_len"], b["aln_start"], c["bam_flags"], b["feats"])
    stripe, rec_off, status, tot = orc.bam_encode(b["ref_id"], b["aln_start"], cg["aln_end"], b["mapq"], c["bam_flags"], c["mate_flags"],
                                                  c["mate_ref"], c["mate_start"], c["tlen"], b["read_len"], b["base_off"], b["bases"],
                                                  b["quals"], np.ones(b["n"], np.uint8), cg["n_cigar"], cg["cigar_off"], cg["cigar"],
                                                  names=b["names"], name_off=b["name_off"], tags=b["tags"], tag_off=b["tag_off"])
    eng = xc.Engine(0)
    try:
        codec = bam.BAMRecordCodec(eng, DEV)
        want = _index_both(codec, orc, stripe)
        assert np.array_equal(want, rec_off)
        _index_both(codec, orc, stripe, cap=max(1, len(want) // 2))   # too small: the count still says what is needed
        _index_both(codec, orc, b"")                                    # an empty stripe
        # decoys: a whole well-formed record (and its successor) as the VALUE of a B:C tag of a long record, placed so that
        # the decoy sits exactly at a segment start
        small = bc.py_bam_record(0, 5, 9, 4681, 0, "d", [(4 << 4)], "ACGT", b"\\5\\5\\5\\5", -1, 0, 0)
        big = bytearray()
        for k in range(40):
            pad = (-(len(big) + 4 + 32 + 3 + 8)) % 4096  # tag bytes: 'XY' 'B' 'C' + count(4) in front of the value
            value = bytes(pad) + small * 3
            tags = b"XYBC" + len(value).to_bytes(4, "little") + value
            big += bc.py_bam_record(1, 100 + k, 30, 4681, 16, "rn", [], "", b"", -1, 0, 0, tags)
            big += small
        want = _index_both(codec, orc, bytes(big))
        assert len(want) - 1 == 80
        # very long records (each spans several segments) between short ones
        long_seq = "ACGT" * 5000
        stripe2 = b"".join(bc.py_bam_record(2, 7 + i, 20, 4681, 0, "long%d" % i, [(20000 << 4)], long_seq, bytes(20000), -1, 0, 0) + small
                           for i in range(6))
        _index_both(codec, orc, stripe2)
        # a block size that runs past the end: CRAMGPU_E_CORRUPT, the records in front of it are listed
        broken = bytearray(stripe2)
        at = int(orc.bam_index(stripe2)[7])
        broken[at:at + 4] = (1 << 28).to_bytes(4, "little")
        d = _t(np.concatenate([np.frombuffer(bytes(broken), np.uint8), np.zeros(16, np.uint8)]))
        off, n, st = codec.index_dev(d, len(broken), 64, 64)
        from cramtools_amd import _lib
        assert st == _lib.E_CORRUPT and n == 7 and np.array_equal(off[:7].cpu().numpy(), orc.bam_index(stripe2)[:7])
    finally:
        eng.close()
