"""The CPU oracle's record kernels against SURVEY.md §8(c) KAT-2/3/4 and the reference's own parser tests."""
import numpy as np

REF = None


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


def test_kat2_features_matrix_restore(katrec, orc):
    ref = katrec["ref"].encode()
    for row in katrec["kat2"]:
        reads = orc.Reads([row["start"]], [row["cigar"]], [row["bases"].encode()])
        f = orc.extract_features(reads, ref)
        assert _feats_of_read(f, 0, row["bases"].encode()) == row["features"], row
        m = orc.substitution_matrix(f["hist"])
        assert bytes(m).hex() == row["matrix"], row
        orc.assign_codes(f, m)
        out, off, st = orc.restore_bases(reads.read_len, reads.aln_start, reads.rflags, f, reads.bases,
                                         reads.base_off, ref, 0, m)
        assert st[0] == 0
        assert out.tobytes().decode() == row["restored"], row


def test_kat4_window_edges(katrec, orc):
    ref = katrec["ref"].encode()
    default = bytes.fromhex("1b1b1b1b1b")
    for row in katrec["kat4"]:
        lo, hi = row["window"]
        fl = row["features"]
        f = dict(op=np.array([ord(x[0]) for x in fl], np.uint8), pos=np.array([x[1] for x in fl], np.int32),
                 a=np.array([(x[2] << 16) if x[0] == "X" else x[2] for x in fl], np.int32),
                 b=np.zeros(len(fl), np.int32), off=np.array([0, len(fl)], np.int64))
        out, off, st = orc.restore_bases([row["len"]], [row["start"]], [0], f, None, None, ref[lo:hi], lo,
                                         np.frombuffer(default, np.uint8))
        assert st[0] == 0
        assert out.tobytes().decode() == row["restored"], row


def test_kat3_lossy(katrec, orc):
    ref = katrec["ref"].encode()
    k = katrec["kat3"]
    reads = orc.Reads([k["start"]], [k["cigar"]], [k["bases"].encode()], quals=[bytes(k["quals"])], mapq=[k["mapq"]])
    f = orc.extract_features(reads, ref)
    assert _feats_of_read(f, 0, k["bases"].encode()) == k["base_features"]
    for row in k["rows"]:
        pols = orc.parse_policies(row["spec"])
        scores, force, status = orc.lossy_quals(reads, ref, pols)
        assert status[0] == 0
        assert int(force[0]) == row["cf"], row
        merged = orc.merge_quality_features(reads, scores, force, status, f)
        rflags = np.array([orc.RF_FORCE_QS if force[0] else 0], np.uint8)
        restored, st = orc.restore_quals(reads.read_len, rflags, merged, scores[:reads.read_len[0]])
        got = [] if st[0] == 1 else list(map(int, restored[:reads.read_len[0]]))
        assert got == row["restored"], row
        if "merged_prefix" in row:
            mf = _feats_of_read(merged, 0, k["bases"].encode())
            for want, have in zip(row["merged_prefix"], mf):
                assert have[:len(want)] == want, (want, have)


def test_parser_reference_tests(katrec, orc):
    """TestQualityScorePreservation.test1/test2 (the only hot-path tests the reference itself runs)."""
    for row in katrec["parser"]:
        pols = orc.parse_policies(row["spec"])
        assert len(pols) == len(row["policies"])
        for got, want in zip(pols, row["policies"]):
            assert got["read_cat"] == want["read_cat"]
            if "read_param" in want:
                assert got["read_param"] == want["read_param"]
            assert [c[0] for c in got["base_cats"]] == want["base_cats"]
            assert all(c[1] == -1 for c in got["base_cats"])
            assert got["treatment"] == want["treatment"] and got["param"] == want["param"]


def test_null_quals_and_tracks_window(orc):
    ref = (b"ACGT" * 600)
    reads = orc.Reads([1, 5], ["4M", "4M"], [b"ACGT", b"ACGA"], quals=[b"\x05" * 4, b"\x05" * 4],
                      rflags=[orc.RF_NULL_QUALS, 0])
    pols = orc.parse_policies("N40")
    scores, force, status = orc.lossy_quals(reads, ref, pols, window=1000)
    assert status[0] == 1 and status[1] == 0
    assert list(scores[4:8]) == [255, 255, 255, 5]
    w = orc.tracks_windows(reads, pols, len(ref), window=1000)
    assert list(w) == [0, 1]
