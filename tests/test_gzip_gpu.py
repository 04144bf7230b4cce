"""gzip / BGZF members written on the device (cramgpu_gzip_dev; SURVEY.md §8(f) rows 2 and 3, encode halves).  DEFLATE
leaves matches and codes to the writer, so the check is not byte identity with zlib (SURVEY §8(f) row 2 puts it outside
the contract) but the property the format defines: zlib's inflater (Python's gzip module — the same RFC 1951 / 1952
reader as java.util.zip.GZIPInputStream) and the gunzip kernel both return the input, the BGZF size fields index the
image, and the members stay within a stated factor of zlib level 5's size (the level htsjdk's writer uses)."""
import gzip
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _itf8ish(rng, n):
    """Bytes shaped like an ITF8 series: mostly one-byte values, some two- and five-byte ones."""
    out = bytearray()
    for v in rng.integers(0, 400, n):
        v = int(v)
        if v < 128:
            out.append(v)
        elif v < 390:
            out += bytes([0x80 | (v >> 8), v & 255])
        else:
            out += bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F])
    return bytes(out)


def _corpus(rng):
    acgt = rng.choice(np.frombuffer(b"ACGT", np.uint8), 70000, p=[0.3, 0.2, 0.2, 0.3]).tobytes()
    quals = rng.integers(33, 74, 50000).astype(np.uint8).tobytes()
    noise = rng.integers(0, 256, 40000).astype(np.uint8).tobytes()
    runs = b"".join(bytes([int(c)]) * int(k) for c, k in zip(rng.integers(0, 256, 300), rng.integers(1, 600, 300)))
    names = b"".join(b"read.%d/%d\t" % (i, i % 7) for i in range(4000))
    flags = rng.choice(np.array([99, 147, 83, 163, 0, 16], np.uint8), 30000, p=[0.3, 0.3, 0.15, 0.15, 0.05, 0.05]).tobytes()
    fib = b"".join(bytes([40 + i]) * f for i, f in enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765]))
    fib = bytes(rng.permutation(np.frombuffer(fib, np.uint8)))  # an optimal code deeper than 15 bits: the limit acts
    return [b"", b"a", b"ab", b"abc", b"abcd", b"abc" * 5, acgt, quals, noise, runs, names, flags, fib, _itf8ish(rng, 20000),
            acgt[:257], bytes(1000), bytes(70000), noise[:5], acgt[:32768], acgt[:32769], names[:65280]]


def test_members_inflate_to_the_input():
    from cramtools_amd import external_compression as xc
    rng = np.random.default_rng(31)
    corpus = _corpus(rng)
    eng = xc.Engine(0)
    try:
        got = eng.gzip_many(corpus, device=DEV)
        for i, (g, d) in enumerate(zip(got, corpus)):
            assert g[:4] == b"\x1f\x8b\x08\x00" and len(g) <= eng.gzip_bound(len(d)), i
            assert gzip.decompress(g) == d, i
            assert struct.unpack("<II", g[-8:]) == (zlib.crc32(d) & 0xFFFFFFFF, len(d)), i
        assert eng.gunzip_many(got, [len(d) for d in corpus], device=DEV) == corpus
        # incompressible input is stored: 5 bytes per 32 KiB block + header and trailer
        assert len(got[8]) <= len(corpus[8]) + 5 * 2 + 18 + 4
        # size beside zlib level 5 (Deflater level htsjdk's GZIP blocks and BGZF use): the compressible entries together
        mine = sum(len(g) for g in got)
        ref = sum(len(gzip.compress(d, 5)) for d in corpus)
        assert mine <= 1.15 * ref, (mine, ref)
        # a second run gives the same bytes (the two parse passes of a block must agree, so must two calls)
        assert eng.gzip_many(corpus, device=DEV) == got
    finally:
        eng.close()


def test_bgzf_image():
    """65280-byte cuts of a payload -> BGZF members in one call; with the EOF marker block behind them the image is what
    BlockCompressedOutputStream writes: gzip readers inflate it, cramgpu_bgzf_index walks its BC fields."""
    from cramtools_amd import _lib, external_compression as xc
    rng = np.random.default_rng(37)
    payload = b"".join(_corpus(rng)) * 2
    chunks = [payload[i:i + 65280] for i in range(0, len(payload), 65280)] + [b""]
    eng = xc.Engine(0)
    try:
        members = eng.gzip_many(chunks, bgzf=True, device=DEV)
        for m in members:
            assert m[:4] == b"\x1f\x8b\x08\x04" and m[10:16] == b"\x06\x00BC\x02\x00" and struct.unpack("<H", m[16:18])[0] == len(m) - 1
        image = b"".join(members)
        assert gzip.decompress(image) == payload
        img = np.frombuffer(image, np.uint8)
        L = _lib.lib()
        n = len(chunks)
        off, usize = np.zeros(n + 1, np.int64), np.zeros(n, np.int32)
        assert _lib.check(L.cramgpu_bgzf_index(img.ctypes.data, img.size, off.ctypes.data, usize.ctypes.data, n + 1)) == n
        assert usize.tolist() == [len(c) for c in chunks] and off[-1] == img.size
        assert b"".join(eng.gunzip_many([image[off[i]:off[i + 1]] for i in range(n)], usize, device=DEV)) == payload
        # 65280 incompressible bytes (htsjdk's block size) still fit a member: stored blocks cost 5 bytes per 32 KiB
        noise = rng.integers(0, 256, 65280).astype(np.uint8).tobytes()
        m = eng.gzip_many([noise], bgzf=True, device=DEV)[0]
        assert len(m) <= 65280 + 18 + 8 + 2 * 5 and gzip.decompress(m) == noise
        # 65536 random bytes cannot fit a 64 KiB member
        with pytest.raises(_lib.CramGpuError) as e:
            eng.gzip_many([rng.integers(0, 256, 65536).astype(np.uint8).tobytes()], bgzf=True, device=DEV)
        assert e.value.code == -2
    finally:
        eng.close()


def test_long_series_as_several_members():
    """A long series cut into pieces, one member each, written back to back: GZIPInputStream-style readers (Python's gzip,
    the gunzip kernel) read the concatenation as ONE stream — the way to spread a long series over many lanes."""
    from cramtools_amd import external_compression as xc
    rng = np.random.default_rng(43)
    series = _itf8ish(rng, 60000)
    pieces = [series[i:i + 8192] for i in range(0, len(series), 8192)]
    eng = xc.Engine(0)
    try:
        image = b"".join(eng.gzip_many(pieces, device=DEV))
        assert gzip.decompress(image) == series
        assert eng.gunzip_many([image], [len(series)], device=DEV) == [series]
    finally:
        eng.close()


def test_slot_too_small():
    from cramtools_amd import _lib, external_compression as xc
    rng = np.random.default_rng(41)
    d = rng.choice(np.frombuffer(b"ACGT", np.uint8), 9000).tobytes()
    eng = xc.Engine(0)
    try:
        size = len(eng.gzip_many([d], device=DEV)[0])
        assert eng.gzip_many([d], device=DEV, caps=[size])[0][-4:] == struct.pack("<I", len(d))
        for cap in (size - 1, 17, 3, 0):
            with pytest.raises(_lib.CramGpuError) as e:
                eng.gzip_many([d], device=DEV, caps=[cap])
            assert e.value.code == -2
        with pytest.raises(_lib.CramGpuError) as e:
            eng.gzip_many([d], device=DEV, caps=[-1])
        assert e.value.code == -6
    finally:
        eng.close()
