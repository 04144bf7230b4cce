"""The data-series oracle (oracle/series.c): ITF8 against the width rule SURVEY.md §8(d) states, the C writer against
an independent plain-Python statement of Appendix A's field order, and writer -> reader round trips.  No fixture of the
reference holds a Writer output: these tests pin the two restatements to each other and to the survey ("parity unpinned"
at the reference level)."""
import numpy as np
import pytest

from tests import series_cases as sc


def test_itf8_width_rule(orc):
    vals = [0, 1, 127, 128, 255, 16383, 16384, (1 << 21) - 1, 1 << 21, (1 << 28) - 1, 1 << 28, 0x7FFFFFFF, -1, -2, -(1 << 31)]
    for v in vals:
        assert orc.itf8([v]) == sc.py_itf8(v), v
    assert orc.itf8([127]) == b"\x7f" and orc.itf8([128]) == b"\x80\x80" and orc.itf8([16384]) == b"\xc0\x40\x00"
    assert orc.itf8([1 << 21]) == b"\xe0\x20\x00\x00" and orc.itf8([1 << 28]) == b"\xf1\x00\x00\x00\x00"
    assert orc.itf8([-1]) == b"\xff\xff\xff\xff\x0f"


@pytest.mark.parametrize("seed", range(6))
def test_writer_matches_python_statement_and_round_trips(orc, seed):
    rng = np.random.default_rng(seed)
    s = sc.random_slice(rng, 120 if seed else 1)
    multiref, ap_delta = bool(seed & 1), seed != 3
    ap_base = int(s["aln_start"][0]) - (seed % 3)
    got = orc.write_series(s["cols"], s["read_len"], s["aln_start"], s["mapq"], s["bases"], s["quals"], s["base_off"], s["feats"],
                           multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
    want = sc.py_write_series(s, multiref, ap_base, ap_delta)
    for k in range(30):
        assert got[k] == want[k], k
    back = orc.read_series(got, s["n"], multiref=multiref, ap_base=ap_base, ap_delta=ap_delta)
    c = s["cols"]
    cf, unm = c["comp_flags"], (c["bam_flags"] & 4) != 0
    assert np.array_equal(back["bam_flags"], c["bam_flags"]) and np.array_equal(back["comp_flags"], cf)
    assert np.array_equal(back["read_len"], s["read_len"]) and np.array_equal(back["aln_start"], s["aln_start"])
    assert np.array_equal(back["read_group"], c["read_group"]) and np.array_equal(back["tag_list"], c["tag_list"])
    if multiref:
        assert np.array_equal(back["ref_id"], c["ref_id"])
    det, down = (cf & 2) != 0, ((cf & 2) == 0) & ((cf & 4) != 0)
    for k in ("mate_flags", "mate_ref", "mate_start", "tlen"):
        assert np.array_equal(back[k][det], c[k][det]) and (back[k][~det] == 0).all()
    assert np.array_equal(back["next_frag"][down], c["next_frag"][down])
    assert np.array_equal(back["mapq"][~unm], s["mapq"][~unm])
    f, n = s["feats"], back["n_feat"]
    assert n == f["op"].size and np.array_equal(back["feat_off"], f["off"])
    assert np.array_equal(back["op"][:n], f["op"]) and np.array_equal(back["pos"][:n], f["pos"])
    for r in range(s["n"]):
        b0, L = int(s["base_off"][r]), int(s["read_len"][r])
        o0 = int(back["base_off"][r])
        if cf[r] & 1:
            assert np.array_equal(back["quals"][o0:o0 + L], s["quals"][b0:b0 + L])
        else:
            assert (back["quals"][o0:o0 + L] == 0xFF).all()
        if unm[r] and not cf[r] & 8:
            assert np.array_equal(back["bases"][o0:o0 + L], s["bases"][b0:b0 + L])
        for k in range(int(f["off"][r]), int(f["off"][r + 1])):
            ch = chr(int(f["op"][k]))
            if ch == "X":
                assert back["a"][k] == f["a"][k] & 0xFF0000
            elif ch in "IS":
                q0 = int(back["seq_off"][r]) + int(back["a"][k])
                assert back["b"][k] == f["b"][k]
                assert np.array_equal(back["seq"][q0:q0 + int(f["b"][k])], s["bases"][b0 + int(f["a"][k]):b0 + int(f["a"][k]) + int(f["b"][k])])
            elif ch == "B":
                assert back["a"][k] == f["a"][k] & 255 and back["b"][k] == f["b"][k] & 255
            elif ch in "iQ":
                assert back["a"][k] == f["a"][k] & 255
            else:
                assert back["a"][k] == f["a"][k]


def test_reader_reports_truncation(orc):
    rng = np.random.default_rng(11)
    s = sc.random_slice(rng, 40)
    got = orc.write_series(s["cols"], s["read_len"], s["aln_start"], s["mapq"], s["bases"], s["quals"], s["base_off"], s["feats"])
    cut = list(got)
    cut[0] = cut[0][:-1]
    with pytest.raises(RuntimeError):
        orc.read_series(cut, s["n"])
