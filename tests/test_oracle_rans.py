"""The CPU oracle's rANS against the golden vectors minted from htsjdk-2.4.1."""
import hashlib

import pytest

from kat_inputs import fuzz_inputs, kat_input


def test_kat1_hex(kat1, orc):
    for row in kat1["hex"]:
        data = kat_input(row["input"])
        out = orc.rans(data, row["order"])
        assert out.hex() == row["out"], row["input"]
        assert orc.unrans(out) == data


def test_kat1_sha256(kat1, orc):
    for row in kat1["sha256"]:
        data = kat_input(row["input"])
        for order, key in ((0, "o0"), (1, "o1")):
            out = orc.rans(data, order)
            assert len(out) == row[key][0], (row["input"], order)
            assert hashlib.sha256(out).hexdigest() == row[key][1], (row["input"], order)
            assert orc.unrans(out) == data


def test_empty_and_short(orc):
    assert orc.rans(b"", 0) == b"" and orc.rans(b"", 1) == b""
    assert orc.unrans(b"") == b""
    for d in (b"a", b"ab", b"abc"):
        # RANS#compress L60-73: fewer than 4 bytes are always coded order 0
        assert orc.rans(d, 1) == orc.rans(d, 0)
        assert orc.rans(d, 1)[0] == 0


def test_bound(orc):
    assert orc.rans_bound(0) == 198151
    assert orc.rans_bound(1000) == int(1.05 * 1000 + 198151)


def test_roundtrip_fuzz(orc):
    for data in fuzz_inputs():
        for order in (0, 1):
            c = orc.rans(data, order)
            assert c[0] == (order if len(data) >= 4 else 0)
            assert int.from_bytes(c[1:5], "little") == len(c) - 9
            assert int.from_bytes(c[5:9], "little") == len(data)
            assert orc.unrans(c) == data


def test_errors(orc):
    c = bytearray(orc.rans(b"abracadabra", 0))
    with pytest.raises(RuntimeError):
        orc.unrans(bytes(c[:-1]))  # "Incorrect input length."
    with pytest.raises(RuntimeError):
        orc.rans(b"abcd", 2)  # "Unknown rANS order"
