"""The container-assembly / BAM oracle (oracle/bam.c) against answers worked by hand from the reference's sources:
CIGARs from features (ReadFeatureBuffer.java:180-294), mate restoration (CramNormalizer.java:82-105, :255-285, :455-470),
one BAM record against an independent plain-Python statement of the SAM specification's layout (the layout
BAMRecordView.java:23-35 writes), and encode -> index -> decode round trips.  No fixture of the reference holds these
outputs ("parity unpinned" at the reference level)."""
import numpy as np
import pytest

from tests import bam_cases as bc

M, I, D, N, S, H, P = range(7)


def _feats(per_read):
    """per_read: list of lists of (op char, pos, a, b)."""
    off = np.concatenate([[0], np.cumsum([len(f) for f in per_read])]).astype(np.int64)
    flat = [x for f in per_read for x in f]
    return dict(off=off, op=np.array([ord(x[0]) for x in flat], np.uint8), pos=np.array([x[1] for x in flat], np.int32),
                a=np.array([x[2] for x in flat], np.int32), b=np.array([x[3] for x in flat], np.int32))


CIGAR_CASES = [
    # (read length, features, expected elements, expected span)
    (100, [], [(100, M)], 100),
    (100, [("X", 50, 0, 0)], [(100, M)], 100),
    (100, [("S", 1, 0, 5)], [(5, S), (95, M)], 95),
    (100, [("D", 51, 3, 0)], [(50, M), (3, D), (50, M)], 103),
    (30, [("I", 11, 10, 2), ("X", 20, 0, 0)], [(10, M), (2, I), (18, M)], 28),
    (10, [("H", 1, 5, 0), ("S", 1, 0, 3)], [(5, H), (3, S), (7, M)], 7),
    (10, [("Q", 5, 30, 0)], [(10, M)], 10),
    (10, [("S", 7, 6, 4)], [(6, M), (4, S)], 6),
    (10, [("i", 3, 65, 0), ("i", 4, 67, 0)], [(2, M), (2, I), (6, M)], 8),
    (20, [("N", 6, 1000, 0), ("B", 9, 65, 30)], [(5, M), (1000, N), (15, M)], 20),
    (8, [("P", 4, 2, 0), ("D", 4, 1, 0)], [(3, M), (2, P), (1, D), (5, M)], 9),
    (5, [("X", 1, 0, 0), ("X", 5, 0, 0)], [(5, M)], 5),
]


def test_cigar_known_answers(orc):
    read_len = np.array([c[0] for c in CIGAR_CASES] + [40], np.int32)
    start = np.arange(100, 100 + len(read_len)).astype(np.int32)
    flags = np.zeros(len(read_len), np.int32)
    flags[-1] = 4  # unmapped: no CIGAR, end 0
    got = orc.cigars(read_len, start, flags, _feats([c[1] for c in CIGAR_CASES] + [[]]))
    for r, (L, _, want, span) in enumerate(CIGAR_CASES):
        o0, k = int(got["cigar_off"][r]), int(got["n_cigar"][r])
        assert [(int(w) >> 4, int(w) & 15) for w in got["cigar"][o0:o0 + k]] == want, r
        assert got["aln_end"][r] == start[r] + span - 1, r
    assert got["n_cigar"][-1] == 0 and got["aln_end"][-1] == 0
    assert got["total"] == sum(len(c[2]) for c in CIGAR_CASES) == got["cigar_off"][-1]
    small = orc.cigars(read_len, start, flags, _feats([c[1] for c in CIGAR_CASES] + [[]]), cigar_cap=3)
    assert small["total"] == got["total"] and not small["cigar"].any()


def _cols(n, **kw):
    c = {k: np.zeros(n, np.int32) for k in bc.sc_cols()}
    for k, v in kw.items():
        c[k] = np.asarray(v, np.int32)
    return c


def test_mate_restoration_known_answers(orc):
    # records: 0 -> 1 (pair, second on the reverse strand), 2 detached, 3 -> 4 -> 5 (three fragments, 5 unmapped), 6 unpaired
    bf = [1 | 64, 1 | 16 | 128, 1, 1, 1 | 16, 1 | 4, 0]
    cf = [4, 0, 2, 4, 4, 0, 4]
    nf = [0, 0, 7, 0, 0, 0, 0]
    cols = _cols(7, bam_flags=bf, comp_flags=cf, next_frag=nf, mate_start=[9] * 7, mate_ref=[8] * 7, tlen=[7] * 7, mate_flags=[3] * 7)
    start = np.array([100, 300, 500, 1000, 1100, 1100, 50], np.int32)
    end = np.array([199, 399, 599, 1099, 1199, 0, 149], np.int32)
    ref = np.array([2, 2, 2, 2, 2, -1, 2], np.int32)
    out, st = orc.restore_mates([0, 7], cols, ref, start, end)
    assert st[0] == 0
    assert out["mate_start"].tolist() == [300, 100, 9, 1100, 0, 1000, 9]       # :279, :283-284 (mate without reference: start 0)
    assert out["mate_ref"].tolist() == [2, 2, 8, 2, -1, 2, 8]
    assert out["mate_flags"].tolist() == [1, 0, 3, 1, 2, 0, 3]                  # :280-281
    assert out["tlen"].tolist() == [300, -300, 7, 0, 7, 0, 7]                   # :455-470: 399 - 100 + 1; an unmapped end: 0
    assert out["next_frag"].tolist() == [0, 0, -1, 0, 0, 0, -1]                 # :85
    # a distance that leaves the container: records.get throws
    cols["next_frag"][4] = 2
    _, st = orc.restore_mates([0, 7], cols, ref, start, end)
    assert st[0] == -9
    # second end in front of the first: negative length (:468)
    cols = _cols(2, bam_flags=[1, 1], comp_flags=[4, 0])
    out, st = orc.restore_mates([0, 2], cols, [0, 0], [500, 100], [599, 199])
    assert out["tlen"].tolist() == [-401, 401]


def test_bam_record_known_answer(orc):
    """The alignment r001/1 of the SAM specification's example (§1.1), field by field."""
    seq, name = "TTAGATAAAGGATACTG", "r001"
    cig = [(8, M), (2, I), (4, M), (1, D), (3, M)]
    words = [(l << 4) | o for l, o in cig]
    qual = bytes(range(20, 20 + len(seq)))
    end = 7 + 8 + 4 + 1 + 3 - 1
    tags = b"NMC\x01"
    want = bc.py_bam_record(0, 7, 30, bc.reg2bin(6, end), 99, name, words, seq, qual, 0, 37, 39, tags)
    bases = np.frombuffer(seq.encode(), np.uint8)
    got, rec_off, status, tot = orc.bam_encode([0], [7], [end], [30], [99 & ~0x20], [1], [0], [37], [39], [len(seq)], [0, len(seq)],
                                               bases, np.frombuffer(qual, np.uint8), [1], [len(words)], [0, len(words)], words,
                                               names=np.frombuffer(name.encode(), np.uint8), name_off=[0, 4],
                                               tags=np.frombuffer(tags, np.uint8), tag_off=[0, 4])
    assert got == want and tot == len(want) and status[0] == 0 and rec_off.tolist() == [0, len(want)]
    assert want[:4] == (len(want) - 4).to_bytes(4, "little") and want[36:41] == b"r001\0"
    assert want[41 + 20:41 + 20 + 9] == bytes([0x88, 0x14, 0x18, 0x11, 0x14, 0x41, 0x81, 0x28, 0x40])  # T=8 A=1 G=4 C=2
    # no scores -> 0xFF x length (BAMRecordView.java:176-180); generated name = decimal of name_base + r (CramNormalizer.java:108-117)
    got, _, _, _ = orc.bam_encode([0], [7], [end], [30], [99 & ~0x20], [1], [0], [37], [39], [len(seq)], [0, len(seq)], bases,
                                  np.frombuffer(qual, np.uint8), [0], [len(words)], [0, len(words)], words, name_base=41)
    assert got == bc.py_bam_record(0, 7, 30, bc.reg2bin(6, end), 99, "41", words, seq, None, 0, 37, 39)
    # unplaced, unpaired, unmapped record: start 0 -> pos -1, mapping quality 0, bin 0, no mate (Cram2SamRecordFactory#create)
    got, _, st, _ = orc.bam_encode([-1], [123], [0], [77], [4], [3], [5], [99], [0], [3], [0, 3], np.frombuffer(b"AC@", np.uint8),
                                   np.frombuffer(b"\1\2\3", np.uint8), [1], [0], [0, 0], np.zeros(1, np.uint32), name_base=0)
    want = bc.py_bam_record(-1, 0, 0, 0, 4, "0", [], "ACN", b"\1\2\3", -1, 0, 0)
    assert got == want and st[0] == -9  # '@' is outside the 4-bit table: BAMRecordView.java:412 throws


@pytest.mark.parametrize("seed", range(4))
def test_encode_index_decode_round_trip(orc, seed):
    rng = np.random.default_rng(40 + seed)
    b = bc.random_batch(rng, [50, 1, 120] if seed else [1])
    c = b["cols"]
    cg = orc.cigars(b["read_len"], b["aln_start"], c["bam_flags"], b["feats"])
    has_q = (rng.random(b["n"]) < 0.8).astype(np.uint8)
    stripe, rec_off, status, tot = orc.bam_encode(b["ref_id"], b["aln_start"], cg["aln_end"], b["mapq"], c["bam_flags"], c["mate_flags"],
                                                  c["mate_ref"], c["mate_start"], c["tlen"], b["read_len"], b["base_off"], b["bases"],
                                                  b["quals"], has_q, cg["n_cigar"], cg["cigar_off"], cg["cigar"], names=b["names"],
                                                  name_off=b["name_off"], tags=b["tags"], tag_off=b["tag_off"])
    assert tot == len(stripe) and not status.any()
    assert np.array_equal(orc.bam_index(stripe), rec_off)
    d = orc.bam_decode(stripe, rec_off)
    assert d["ret"] == int(b["base_off"][-1]) and not d["status"].any()
    placed = b["ref_id"] != -1
    paired = (c["bam_flags"] & 1) != 0
    assert np.array_equal(d["ref_id"], b["ref_id"])
    assert np.array_equal(d["aln_start"], np.where(placed, b["aln_start"], 0))
    assert np.array_equal(d["mapq"], np.where(placed, b["mapq"], 0))
    want_flags = (c["bam_flags"] & ~0x28) | np.where(paired & ((c["mate_flags"] & 1) != 0), 0x20, 0) | \
        np.where(paired & ((c["mate_flags"] & 2) != 0), 0x8, 0)
    assert np.array_equal(d["bam_flags"], want_flags & 0xFFFF)
    assert np.array_equal(d["mate_ref"], np.where(paired, c["mate_ref"], -1))
    assert np.array_equal(d["mate_start"], np.where(paired & (c["mate_start"] > 0), c["mate_start"], 0))
    assert np.array_equal(d["tlen"], c["tlen"]) and np.array_equal(d["read_len"], b["read_len"])
    assert np.array_equal(d["base_off"], b["base_off"]) and np.array_equal(d["n_cigar"], cg["n_cigar"])
    assert np.array_equal(d["cigar"][:cg["total"]], cg["cigar"][:cg["total"]])
    nb = int(b["base_off"][-1])
    assert np.array_equal(d["bases"][:nb], b["bases"][:nb])
    for r in range(b["n"]):
        o0, L = int(b["base_off"][r]), int(b["read_len"][r])
        q = d["quals"][o0:o0 + L]
        assert np.array_equal(q, b["quals"][o0:o0 + L]) if has_q[r] else (q == 0xFF).all()
        assert stripe[int(d["name_off"][r]):int(d["name_off"][r]) + int(d["name_len"][r])] == \
            b["names"][int(b["name_off"][r]):int(b["name_off"][r + 1])].tobytes()
        assert stripe[int(d["tag_off"][r]):int(d["tag_off"][r]) + int(d["tag_len"][r])] == \
            b["tags"][int(b["tag_off"][r]):int(b["tag_off"][r + 1])].tobytes()
        want_rf = (1 if c["bam_flags"][r] & 4 else 0) | (2 if (not has_q[r] or b["quals"][o0] == 0xFF) else 0)
        assert d["rflags"][r] == want_rf, r
    # a truncated stripe and a block size below 32 are refused (Bam_OBA_Supplier.java:59-62)
    with pytest.raises(RuntimeError):
        orc.bam_index(stripe[:-1])
    with pytest.raises(RuntimeError):
        orc.bam_index((8).to_bytes(4, "little") + bytes(8))


def test_mates_random_chains_are_consistent(orc):
    """Properties of CramNormalizer.java:255-285 on random containers: a two-record chain mirrors its ends."""
    rng = np.random.default_rng(9)
    b = bc.random_batch(rng, [200, 300])
    c = b["cols"]
    cg = orc.cigars(b["read_len"], b["aln_start"], c["bam_flags"], b["feats"])
    out, st = orc.restore_mates(b["cont_first"], c, b["ref_id"], b["aln_start"], cg["aln_end"])
    assert set(st.tolist()) <= {0, -9}
    linked = ((c["bam_flags"] & 1) != 0) & ((c["comp_flags"] & 2) == 0)
    assert (out["next_frag"][~linked] == -1).all()


def test_set_mate_info_known_answers(orc):
    """CramSerilization.java:274-373 worked by hand.  Records (name, flags):
    0 a 0x41 fwd  + 1 a 0x91 rev: a proper pair (template 399 - 100 + 1 = 300)      -> 0: mate downstream, NF 0; 1: neither bit
    2 b 0x00 not multi-fragment                                                     -> detached (:278-284)
    3 c 0x41 name seen once                                                         -> detached (:350-358)
    4 d 0x41 + 6 d 0x91: template size of 4 off by one                              -> both detached (:334-336)
    5 e, 7 e, 8 e: a third record of the name                                        -> all detached (:294-298)
    9 a 0x141 secondary: its own map (:287), alone there                             -> detached, pair 0/1 untouched
    10 f 0x41 + 13 f 0x81 fwd/fwd, 13 in front of 10: template -(2000 - 1500) - 1   -> 10: mate downstream, NF 2
    11 g 0x41 + 12 g 0x81: mate start of 12 does not name 11's start                 -> both detached (:341-346)"""
    names = [b"a", b"a", b"b", b"c", b"d", b"e", b"d", b"e", b"e", b"a", b"f", b"g", b"g", b"f"]
    bf = [0x41, 0x91, 0x00, 0x41, 0x41, 0x41, 0x91, 0x81, 0x41, 0x141, 0x41, 0x41, 0x81, 0x81]
    start = [100, 300, 10, 20, 100, 700, 300, 800, 900, 100, 2000, 50, 60, 1500]
    end = [199, 399, 19, 29, 199, 799, 399, 899, 999, 199, 2099, 149, 159, 1599]
    n = len(names)
    mate_start = [300, 100, 0, 0, 300, 0, 100, 0, 0, 300, 1500, 60, 51, 2000]
    mate_flags = [1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0]
    tlen = [300, -300, 0, 0, 301, 0, -300, 0, 0, 300, -501, 11, -11, 501]
    cols = _cols(n, bam_flags=bf, comp_flags=[3, 3, 1, 0, 6, 0, 0, 4, 0, 0, 2, 0, 0, 7], next_frag=[9] * n, mate_start=mate_start,
                 mate_flags=mate_flags, tlen=tlen, mate_ref=[0] * n)
    off = np.concatenate([[0], np.cumsum([len(x) for x in names])])
    cf, nf = orc.set_mate_info([0, n], cols, [0] * n, start, end, b"".join(names), off)
    D, MD = 2, 4
    assert cf.tolist() == [1 | MD, 1, 1 | D, D, D, D, D, D, D, D, MD, D, D, 1]
    assert nf.tolist() == [0, 9, -1, -1, -1, -1, -1, -1, -1, -1, 2, -1, -1, 9]
    # the same records as two containers: names do not link across the boundary (each container has its own maps)
    cf2, nf2 = orc.set_mate_info([0, 1, n], cols, [0] * n, start, end, b"".join(names), off)
    assert cf2[0] == 1 | D and cf2[1] == 1 | D and nf2[0] == -1 and cf2.tolist()[2:] == cf.tolist()[2:]


def test_set_mate_info_random_batches_follow_their_construction(orc):
    rng = np.random.default_rng(77)
    b = bc.mate_batch(rng, [1, 2, 60, 500])
    c = b["cols"]
    cf, nf = orc.set_mate_info(b["cont_first"], c, c["ref_id"], b["aln_start"], b["aln_end"], b["names"], b["name_off"])
    for r, k in enumerate(b["kind"]):
        if k in ("single", "lone", "bad", "three", "four"):
            assert cf[r] & 2 and not cf[r] & 4 and nf[r] == -1, (r, k)
            assert (cf[r] & ~6) == (c["comp_flags"][r] & ~6)
        else:  # pair / sec: one record of the two has the mate downstream, neither is detached
            assert not cf[r] & 2, (r, k)
    md = np.flatnonzero(cf & 4)
    assert len(md) == sum(k in ("pair", "sec") for k in b["kind"]) // 2
    for r in md:  # the distance names the record of the same name
        z = r + 1 + nf[r]
        nm = lambda i: bytes(b["names"][b["name_off"][i]:b["name_off"][i + 1]])
        assert nm(r) == nm(z) and (c["bam_flags"][r] & 0x100) == (c["bam_flags"][z] & 0x100)


def test_content_digests_known_answers(orc):
    """CRC-32 check value (ISO 3309 / zlib): crc32("123456789") = 0xCBF43926; the sum wraps at 32 bits."""
    bases = np.frombuffer(b"123456789" + b"123456789" + b"ACGT", np.uint8)
    quals = np.frombuffer(b"!" * 9 + b"123456789" + b"IIII", np.uint8)
    out = orc.content_digests([0, 2, 3, 3], [0, 9, 18], [9, 9, 4], bases, quals)
    import zlib
    assert out[0, 0] == (2 * 0xCBF43926) & 0xFFFFFFFF
    assert out[0, 1] == (zlib.crc32(b"!" * 9) + 0xCBF43926) & 0xFFFFFFFF
    assert out[1, 0] == zlib.crc32(b"ACGT") and out[2].tolist() == [0, 0]


def test_reference_bam_record_test_inputs(orc):
    """The records the reference's own tests build (TestBAMRecordView.java:47-66, BAMRecordViewTest.java:78-124), through the
    oracle's encoder and decoder; expected bytes from an independent statement of BAMRecordView's layout (tests/bam_cases.py)."""
    r = bc.REF_VIEW_RECORD
    bases = np.frombuffer(r["seq"].encode(), np.uint8)
    # flags 555 = paired | proper | mate unmapped (0x8) | mate reverse (0x20) | QC fail: the two mate bits travel as mate flags
    got, rec_off, status, tot = orc.bam_encode([0], [77], [86], [44], [555 & ~0x28], [3], [0], [78], [133], [10], [0, 10], bases,
                                               np.frombuffer(r["qual"], np.uint8), [1], [1], [0, 1], r["cigar"],
                                               names=np.frombuffer(b"name1", np.uint8), name_off=[0, 5],
                                               tags=np.frombuffer(r["tags"], np.uint8), tag_off=[0, 4])
    want = bc.ref_view_record_bytes(bc.reg2bin(76, 86))
    assert got == want and status[0] == 0 and tot == len(want) == 4 + 32 + 6 + 4 + 5 + 10 + 4
    # the bytes as the reference's test makes them (bin 99) read back to the fields the test set
    raw = bc.ref_view_record_bytes(99)
    assert raw[14:16] == (99).to_bytes(2, "little")
    d = orc.bam_decode(raw, orc.bam_index(raw))
    assert (int(d["ref_id"][0]), int(d["aln_start"][0]), int(d["mapq"][0]), int(d["bam_flags"][0])) == (0, 77, 44, 555)
    assert (int(d["mate_ref"][0]), int(d["mate_start"][0]), int(d["tlen"][0]), int(d["read_len"][0])) == (0, 78, 133, 10)
    assert d["bases"][:10].tobytes() == b"A" * 10 and d["quals"][:10].tobytes() == b"B" * 10 and d["cigar"][0] == (10 << 4)
    assert raw[int(d["name_off"][0]):int(d["name_off"][0]) + int(d["name_len"][0])] == b"name1"
    assert raw[int(d["tag_off"][0]):int(d["tag_off"][0]) + int(d["tag_len"][0])] == b"AMAQ"
    # the unmapped record, twice, in one stripe
    names = b"readName2"
    got, rec_off, status, tot = orc.bam_encode([-1, -1], [0, 0], [0, 0], [0, 0], [4, 4], [0, 0], [-1, -1], [0, 0], [0, 0], [1, 1],
                                               [0, 1, 2], np.frombuffer(b"AA", np.uint8), np.zeros(2, np.uint8), [1, 1], [0, 0],
                                               [0, 0, 0], np.zeros(1, np.uint32), names=np.frombuffer(names, np.uint8),
                                               name_off=[0, 8, 9])
    want = bc.ref_unmapped_pair_bytes()
    assert got == want and not status.any() and rec_off.tolist() == [0, 4 + 32 + 9 + 1 + 1, len(want)]
    d = orc.bam_decode(want, orc.bam_index(want))
    assert d["ref_id"].tolist() == [-1, -1] and d["aln_start"].tolist() == [0, 0] and d["bam_flags"].tolist() == [4, 4]
    assert d["read_len"].tolist() == [1, 1] and d["n_cigar"].tolist() == [0, 0] and d["quals"][:2].tolist() == [0, 0]
