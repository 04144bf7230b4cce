"""gunzip on the device (cramgpu_gunzip_dev; SURVEY.md §8(f) rows 2 and 3, decode halves) against zlib: stored, fixed and
dynamic DEFLATE blocks at every compression level, several members in a row, headers with name / extra / comment fields,
BGZF blocks indexed by cramgpu_bgzf_index, empty members, and the failures GZIPInputStream reports (bad magic, truncated
data, CRC-32 / ISIZE mismatch, invalid block type, output larger than the block header's raw size)."""
import gzip
import io
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=None, extra=None, comment=None, hcrc=None):
    """One RFC 1952 member built by hand around zlib's raw DEFLATE."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = co.compress(data) + co.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    hdr = struct.pack("<BBBBIBB", 0x1F, 0x8B, 8, flg, 0, 0, 255)
    if extra is not None:
        hdr += struct.pack("<H", len(extra)) + extra
    if name is not None:
        hdr += name + b"\0"
    if comment is not None:
        hdr += comment + b"\0"
    if hcrc:  # FHCRC: low 16 bits of the CRC-32 of the header ("bad": one off)
        hdr += struct.pack("<H", ((zlib.crc32(hdr) & 0xFFFF) + (1 if hcrc == "bad" else 0)) & 0xFFFF)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def _bgzf_block(data, level=5):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    bsize = 18 + len(body) + 8
    return (struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _corpus(rng):
    acgt = rng.choice(np.frombuffer(b"ACGT", np.uint8), 70000, p=[0.3, 0.2, 0.2, 0.3]).tobytes()
    quals = rng.integers(33, 74, 50000).astype(np.uint8).tobytes()
    noise = rng.integers(0, 256, 40000).astype(np.uint8).tobytes()
    runs = b"".join(bytes([int(c)]) * int(k) for c, k in zip(rng.integers(0, 256, 300), rng.integers(1, 600, 300)))
    names = b"".join(b"read.%d/%d\t" % (i, i % 7) for i in range(4000))
    # matches of every length 3..258 (the copy works in 16-byte pieces with a tail of 0..15 bytes), far and near: periods
    # around the 16-byte limit below which the copy goes in 4-byte turns
    seed = noise[:600]
    every_len = seed + b"".join(seed[L % 37:L % 37 + L] + bytes([L & 255, (L * 7) & 255]) for L in range(3, 259))
    periods = b"".join(seed[k:2 * k] * (3 + k % 5) + seed[:k % 7] for k in range(4, 40))
    return [b"", b"a", b"abc" * 5, acgt, quals, noise, runs, names, acgt[:257], bytes(1000), noise[:5], every_len, periods]


# both kernels: a wave per stream ("64": what a launch of few streams gets) and a lane per stream ("2": a launch of many)
KERNELS = ("64", "2")


@pytest.mark.parametrize("kernel", KERNELS)
def test_gunzip_matches_zlib(kernel, monkeypatch):
    from cramtools_amd import external_compression as xc
    monkeypatch.setenv("CRAMGPU_GUNZIP_LANES", kernel)
    rng = np.random.default_rng(17)
    corpus = _corpus(rng)
    streams, want = [], []
    for i, d in enumerate(corpus):
        for level in (0, 1, 5, 9):
            streams.append(_member(d, level))
            want.append(d)
        streams.append(_member(d, 6, zlib.Z_FIXED))
        want.append(d)
        streams.append(_member(d, 6, zlib.Z_HUFFMAN_ONLY, name=b"series.bin", extra=b"xy\x03\x00abc", comment=b"c"))
        want.append(d)
        streams.append(_member(d, 5, name=b"n", hcrc=True))
        want.append(d)
        streams.append(gzip.compress(d, 5))  # what java.util.zip.GZIPOutputStream-like writers produce
        want.append(d)
    # several members in a row are one stream (GZIPInputStream reads through them); trailing bytes that are no header are ignored
    streams.append(_member(corpus[3], 5) + _member(corpus[4], 1) + _member(b"", 9) + _member(corpus[7], 9))
    want.append(corpus[3] + corpus[4] + corpus[7])
    streams.append(_member(corpus[6], 5) + b"\0\0\0\0\0")
    want.append(corpus[6])
    for s, w in zip(streams, want):
        assert gzip.decompress(s) == w or s.endswith(b"\0\0\0\0\0")
    eng = xc.Engine(0)
    try:
        got = eng.gunzip_many(streams, [len(w) for w in want], device=DEV)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, i
    finally:
        eng.close()


def test_wave_kernel_small_streams(monkeypatch):
    """The wave-per-stream kernel on streams small enough for the thread-per-lane host build (tests/test_glue_hostsim.py):
    stored, fixed and dynamic blocks, overlapping runs, matches longer than 64 bytes, two members in a row, an empty one."""
    from cramtools_amd import external_compression as xc
    monkeypatch.setenv("CRAMGPU_GUNZIP_LANES", "64")
    rng = np.random.default_rng(5)
    text = b"".join(b"read.%d/%d\t" % (i, i % 7) for i in range(60))
    seed = rng.integers(0, 256, 200).astype(np.uint8).tobytes()
    datas = [b"", b"a", b"abcabcabcabcabcabcabcabc", text, seed + seed[:150] + seed[20:190], bytes(700) + b"\1\2" * 300,
             seed[:40]]
    streams, want = [], []
    for i, d in enumerate(datas):
        for level in (0, 1, 6):
            streams.append(_member(d, level))
            want.append(d)
    streams.append(_member(text, 6, zlib.Z_FIXED))
    want.append(text)
    streams.append(_member(text, 6) + _member(seed, 1) + b"\0\0\0")  # two members and padding behind them
    want.append(text + seed)
    eng = xc.Engine(0)
    try:
        got = eng.gunzip_many(streams, [len(w) for w in want], device=DEV)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, i
        bad = bytearray(_member(text, 6))
        bad[-6] ^= 1  # CRC-32
        with pytest.raises(Exception):
            eng.gunzip_many([bytes(bad)], [len(text)], device=DEV)
        with pytest.raises(Exception):
            eng.gunzip_many([_member(text, 6)[:-20]], [len(text)], device=DEV)  # truncated
    finally:
        eng.close()


def test_bgzf_blocks():
    """A BAM-like BGZF image: indexed on the host from the BC subfields, every block inflated in one call."""
    import torch
    from cramtools_amd import _lib, external_compression as xc
    rng = np.random.default_rng(23)
    payload = b"".join(_corpus(rng)) * 3
    chunks = [payload[i:i + 65280] for i in range(0, len(payload), 65280)]
    image = b"".join(_bgzf_block(c) for c in chunks) + _bgzf_block(b"")  # the EOF marker block
    assert gzip.decompress(image) == payload
    img = np.frombuffer(image, np.uint8)
    L = _lib.lib()
    n = _lib.check(L.cramgpu_bgzf_index(img.ctypes.data, img.size, None, None, 0))
    assert n == len(chunks) + 1
    off, usize = np.zeros(n + 1, np.int64), np.zeros(n, np.int32)
    assert _lib.check(L.cramgpu_bgzf_index(img.ctypes.data, img.size, off.ctypes.data, usize.ctypes.data, n + 1)) == n
    assert usize.tolist() == [len(c) for c in chunks] + [0] and off[-1] == img.size
    with pytest.raises(_lib.CramGpuError):
        _lib.check(L.cramgpu_bgzf_index(img.ctypes.data, img.size - 1, None, None, 0))
    eng = xc.Engine(0)
    try:
        got = eng.gunzip_many([image[off[i]:off[i + 1]] for i in range(n)], usize, device=DEV)
        assert b"".join(got) == payload
    finally:
        eng.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_gunzip_failures(kernel, monkeypatch):
    monkeypatch.setenv("CRAMGPU_GUNZIP_LANES", kernel)
    import torch
    from cramtools_amd import _lib, external_compression as xc
    rng = np.random.default_rng(29)
    data = rng.choice(np.frombuffer(b"ACGT", np.uint8), 5000).tobytes()
    good = _member(data, 5)
    bad_crc = bytearray(good); bad_crc[-8] ^= 1
    bad_len = bytearray(good); bad_len[-1] ^= 1
    bad_magic = bytearray(good); bad_magic[1] = 0x8C
    bad_type = struct.pack("<BBBBIBB", 0x1F, 0x8B, 8, 0, 0, 0, 255) + b"\x07" + bytes(20)  # BTYPE = 3
    flipped = bytearray(good); flipped[40] ^= 0x55
    cases = [(bytes(bad_crc), len(data), -5), (bytes(bad_len), len(data), -5), (bytes(bad_magic), len(data), -5),
             (good[:-9], len(data), -5), (good[:60], len(data), -5), (bad_type, 100, -5), (b"", 10, -5), (good, len(data) - 1, -2),
             (good, len(data), len(data)),
             (good + good[:30], len(data), -5), (_member(data, 5, hcrc="bad"), len(data), -5),
             (good + _member(data, 5, hcrc="bad"), 2 * len(data), len(data))]  # a bad header behind a good member ends the stream quietly  # a second member that opens with a valid header and then ends: EOFException
    try:
        zlib.decompress(bytes(flipped), 31)
        flipped_ok = True
    except zlib.error:
        flipped_ok = False
    eng = xc.Engine(0)
    try:
        n = len(cases) + 1
        blobs = [c[0] for c in cases] + [bytes(flipped)]
        caps = [c[1] for c in cases] + [len(data) + 64]
        desc = (_lib.Block * n)()
        io_, oo = 0, 0
        for i in range(n):
            desc[i] = _lib.Block(io_, oo, len(blobs[i]), caps[i], 0, 0)
            io_ += len(blobs[i]); oo += caps[i]
        d_in = torch.as_tensor(np.frombuffer(b"".join(blobs) + bytes(16), np.uint8).copy()).to(DEV)
        d_out = torch.zeros(oo + 16, dtype=torch.uint8, device=DEV)
        d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(DEV)
        d_len = torch.zeros(n, dtype=torch.int32, device=DEV)
        eng.gunzip_dev(d_in, d_out, d_desc, n, d_len)
        lens = d_len.cpu().numpy()
        assert lens[:len(cases)].tolist() == [c[2] for c in cases]
        assert (lens[-1] >= 0) == flipped_ok  # a flipped bit inside the data: refused exactly when zlib refuses it
        out = d_out.cpu().numpy()
        assert out[desc[8].out_off:desc[8].out_off + len(data)].tobytes() == data
    finally:
        eng.close()
