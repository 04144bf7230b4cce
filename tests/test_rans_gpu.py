"""Parity of the HIP rANS codec (through the C ABI) with the golden vectors and with the CPU oracle."""
import hashlib

import numpy as np
import pytest

from kat_inputs import fuzz_inputs, kat_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xc():
    from cramtools_amd import external_compression as x
    return x


@pytest.fixture(scope="module")
def engine(xc):
    e = xc.Engine(0)
    yield e
    e.close()


def test_kat1_hex(kat1, xc):
    for row in kat1["hex"]:
        data = kat_input(row["input"])
        out = xc.rans(data, row["order"])
        assert out.hex() == row["out"], row["input"]
        assert xc.unrans(out) == data


def test_kat1_sha256(kat1, engine, xc):
    datas, orders, want = [], [], []
    for row in kat1["sha256"]:
        d = kat_input(row["input"])
        for order, key in ((0, "o0"), (1, "o1")):
            datas.append(d)
            orders.append(order)
            want.append(row[key])
    outs = engine.compress_many(datas, orders)
    for d, o, w in zip(datas, outs, want):
        assert len(o) == w[0]
        assert hashlib.sha256(o).hexdigest() == w[1]
    back = engine.uncompress_many(outs)
    assert back == datas


def test_empty_short_and_errors(xc, orc):
    assert xc.rans(b"", 0) == b"" and xc.rans(b"", 1) == b"" and xc.unrans(b"") == b""
    for d in (b"a", b"ab", b"abc"):
        assert xc.rans(d, 1) == orc.rans(d, 0)
    with pytest.raises(RuntimeError):
        xc.rans(b"abcd", 2)
    c = xc.rans(b"abracadabra", 0)
    with pytest.raises(RuntimeError):
        xc.unrans(c[:-1])  # "Incorrect input length."
    bad = bytearray(c)
    bad[0] = 7
    with pytest.raises(RuntimeError):
        xc.unrans(bytes(bad))


def test_fuzz_vs_oracle(engine, orc):
    datas = fuzz_inputs()
    rng = np.random.default_rng(5)
    # wide alphabets (order-1 tables in global scratch), skewed and run-heavy inputs, sizes around the batch size
    for n in (64, 65, 127, 1023, 4097, 70000):
        datas.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        datas.append(rng.integers(0, 60, n, dtype=np.uint8).tobytes())
        datas.append(np.repeat(rng.integers(0, 7, n // 5 + 1, dtype=np.uint8), 5)[:n].tobytes())
        datas.append((rng.geometric(0.3, n) % 256).astype(np.uint8).tobytes())
    for order in (0, 1):
        outs = engine.compress_many(datas, [order] * len(datas))
        for d, o in zip(datas, outs):
            assert o == orc.rans(d, order), (len(d), order)
        assert engine.uncompress_many(outs) == datas


def test_decoder_accepts_oracle_streams(engine, orc):
    rng = np.random.default_rng(9)
    datas = [rng.integers(0, k, n, dtype=np.uint8).tobytes() for k in (2, 5, 41, 200, 256) for n in (5, 333, 20000)]
    for order in (0, 1):
        streams = [orc.rans(d, order) for d in datas]
        assert engine.uncompress_many(streams) == datas


def test_large_quality_block(engine, orc):
    from cramtools_amd import synth
    q = synth.markov_quals(1500000, 2).tobytes()
    outs = engine.compress_many([q, q], [0, 1])
    assert outs[0] == orc.rans(q, 0) and outs[1] == orc.rans(q, 1)
    assert engine.uncompress_many(outs) == [q, q]


def test_corrupt_streams_do_not_crash(engine, orc):
    d = bytes(range(64)) * 20
    for order in (0, 1):
        c = bytearray(orc.rans(d, order))
        for cut in (9, 12, 40):
            t = bytes(c[:cut])
            t = t[:1] + (len(t) - 9).to_bytes(4, "little") + t[5:]
            with pytest.raises(RuntimeError):
                engine.uncompress_many([t])


def test_async_jobs(engine, orc):
    """cramgpu_rans_submit / poll / wait: several containers' worth of blocks queued before the first wait give what
    the synchronous batch gives; a fifth outstanding job is refused; decode jobs round-trip."""
    rng = np.random.default_rng(3)
    batches = []
    for k in range(4):
        datas = [rng.integers(0, 5 + 40 * k, n, dtype=np.uint8).tobytes() for n in (30000 + k, 777, 5, 12000)]
        batches.append((datas, [1, 0, 1, 1]))
    jobs = [engine.submit_compress(d, o) for d, o in batches]
    from cramtools_amd._lib import CramGpuError
    with pytest.raises(CramGpuError):
        engine.submit_compress(*batches[0])
    outs = []
    for (d, o), j in zip(batches, jobs):
        got = j.wait()
        assert got == [orc.rans(x, y) for x, y in zip(d, o)]
        outs.append(got)
    djobs = [engine.submit_uncompress(o) for o in outs]
    while not djobs[0].done():
        pass
    for (d, _), j in zip(batches, djobs):
        assert j.wait() == d
    assert engine.compress_many(*batches[1]) == outs[1]  # the synchronous path still works afterwards
