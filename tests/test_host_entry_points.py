"""Host-only C-ABI entry points (no GPU needed): policy parser, ReferenceTracks window replay, substitution matrix."""
import ctypes as C

import numpy as np

from cramtools_amd import _lib

TREAT = {0: "PRESERVE", 1: "BIN", 2: "DROP"}
READ = {0: None, 1: "UNPLACED", 2: "HIGHER_MAPPING_SCORE", 3: "LOWER_MAPPING_SCORE", 4: "ALL"}
BASE = {1: "MATCH", 2: "MISMATCH", 3: "FLANKING_DELETION", 4: "PILEUP", 5: "LOWER_COVERAGE", 6: "INSERTION"}


def _parse(spec):
    arr = (_lib.Policy * 32)()
    n = _lib.lib().cramgpu_parse_policies(spec.encode(), arr, 32)
    assert n >= 0, spec
    return [arr[i] for i in range(n)]


def test_parser_reference_tests(katrec):
    """TestQualityScorePreservation.test1/test2 (src/test/java/htsjdk/samtools/cram/lossy/...:46-114)."""
    for row in katrec["parser"]:
        pols = _parse(row["spec"])
        assert len(pols) == len(row["policies"])
        for got, want in zip(pols, row["policies"]):
            assert READ[got.read_cat] == want["read_cat"]
            if "read_param" in want:
                assert got.read_param == want["read_param"]
            assert [BASE[got.base_cat[k]] for k in range(got.n_base_cat)] == want["base_cats"]
            assert TREAT[got.treatment] == want["treatment"]


def test_parser_matches_oracle(orc):
    specs = ["N40-D8", "R8-N40-D8", "*8", "m5", "I40", "R8", "R8-N40-D8-X0", "R8-N40-D8-P9999", "U0-M30_8", "N0-R40-D0",
             "X5", "P3", "_8", "R40X5", "m999_8"]
    for spec in specs:
        got = _parse(spec)
        want = orc.parse_policies(spec)
        assert len(got) == len(want), spec
        for g, w in zip(got, want):
            assert g.read_cat == orc.READ_CATS[w["read_cat"]], spec
            assert g.treatment == orc.TREATMENTS[w["treatment"]], spec
            assert [(g.base_cat[k], g.base_param[k]) for k in range(g.n_base_cat)] == \
                [(orc.BASE_CATS[a], b) for a, b in w["base_cats"]], spec
            if w["read_cat"] in ("HIGHER_MAPPING_SCORE", "LOWER_MAPPING_SCORE"):
                assert g.read_param == w["read_param"], spec
    arr = (_lib.Policy * 4)()
    assert _lib.lib().cramgpu_parse_policies(b"Z8", arr, 4) == _lib.E_ARG  # "Unknown read or base category"


def test_tracks_windows_matches_oracle(orc):
    rng = np.random.default_rng(1)
    n = 400
    starts = np.sort(rng.integers(1, 5000, n)).astype(np.int32)
    starts[rng.integers(0, n, 5)] = 0
    span = rng.integers(10, 300, n).astype(np.int32)
    rflags = np.where(rng.random(n) < 0.05, 2, 0).astype(np.uint8)
    mapq = rng.integers(0, 60, n).astype(np.uint8)

    class R:
        pass
    reads = R()
    reads.n, reads.aln_start, reads.aln_span, reads.rflags, reads.mapq = n, starts, span, rflags, mapq
    for spec in ("N40-D8", "m5", "*8", "M30N8"):
        pols = orc.parse_policies(spec)
        arr = (_lib.Policy * 32)()
        npol = _lib.lib().cramgpu_parse_policies(spec.encode(), arr, 32)
        for window in (1000000, 900):
            want = orc.tracks_windows(reads, pols, 5400, window)
            got = np.zeros(n, np.int32)
            rc = _lib.lib().cramgpu_tracks_windows(n, starts.ctypes.data, span.ctypes.data, rflags.ctypes.data,
                                                   mapq.ctypes.data, arr, npol, 5400, window, got.ctypes.data, None)
            assert rc == 0
            assert np.array_equal(got, want), (spec, window)
    # an unsorted container makes ReferenceTracks throw ("Cannot move the window backwards")
    bad = starts.copy()
    bad[10], bad[300] = bad[300], bad[10]
    arr = (_lib.Policy * 32)()
    npol = _lib.lib().cramgpu_parse_policies(b"N40", arr, 32)
    got = np.zeros(n, np.int32)
    import ctypes
    at = ctypes.c_int32(-1)
    assert _lib.lib().cramgpu_tracks_windows(n, bad.ctypes.data, span.ctypes.data, rflags.ctypes.data, mapq.ctypes.data,
                                             arr, npol, 5400, 900, got.ctypes.data, ctypes.byref(at)) == _lib.E_JAVA_THROWS
    want_rc = orc.tracks_windows_rc(reads_with(bad, reads), orc.parse_policies("N40"), 5400, 900)
    assert want_rc == -(at.value + 1)
    # terms that read the coverage / mismatch tracks are refused; their no-op spellings are not
    for spec, rc_want in ((b"R8-X5", _lib.E_UNSUPPORTED), (b"P3", _lib.E_UNSUPPORTED), (b"R8-N40-D8-X0", 0),
                          (b"R8-N40-D8-P9999", 0)):
        npol = _lib.lib().cramgpu_parse_policies(spec, arr, 32)
        assert _lib.lib().cramgpu_tracks_windows(n, starts.ctypes.data, span.ctypes.data, rflags.ctypes.data,
                                                 mapq.ctypes.data, arr, npol, 5400, 900, got.ctypes.data, None) == rc_want


def reads_with(starts, like):
    class R:
        pass
    r = R()
    r.n, r.aln_start, r.aln_span, r.rflags, r.mapq = like.n, starts, like.aln_span, like.rflags, like.mapq
    return r


def test_substitution_matrix_matches_oracle_and_kat(orc, katrec):
    rng = np.random.default_rng(2)
    for _ in range(200):
        h = rng.integers(0, 4, 25).astype(np.int64) * rng.integers(0, 3)
        m = np.zeros(5, np.uint8)
        _lib.lib().cramgpu_substitution_matrix(h.ctypes.data, m.ctypes.data)
        assert np.array_equal(m, orc.substitution_matrix(h))
    z = np.zeros(25, np.int64)
    m = np.zeros(5, np.uint8)
    _lib.lib().cramgpu_substitution_matrix(z.ctypes.data, m.ctypes.data)
    assert bytes(m).hex() == "1b1b1b1b1b"  # the default matrix of KAT-2
