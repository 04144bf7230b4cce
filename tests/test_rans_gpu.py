"""Parity of the HIP rANS codec (through the C ABI) with the golden vectors and with the CPU oracle."""
import hashlib

import numpy as np
import pytest

from kat_inputs import fuzz_inputs, kat_input

pytestmark = pytest.mark.gpu

DEV = "cuda:0"  # the host-build glue test (tests/test_glue_hostsim.py) swaps in "cpu"


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
    with pytest.raises(RuntimeError, match=r"Unknown rANS order: 2$"):  # RANS.ORDER.fromInt's message, X filled in
        xc.rans(b"abcd", 2)
    c = xc.rans(b"abracadabra", 0)
    with pytest.raises(RuntimeError, match=r"Incorrect input length\."):
        xc.unrans(c[:-1])
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


def test_order1_narrow_tables(engine, orc):
    """Dense alphabets of 2..16 symbols decode through the hand-scheduled narrow path (rans_dec.hip, narrow_block):
    both table shapes (<= 8 and 9..16 symbols), sizes around the 256-step blocks, skewed inputs whose rare symbols take
    two renormalisation bytes, and streams cut or flipped inside the payload (no crash, no write outside the block)."""
    rng = np.random.default_rng(77)
    datas = []
    for nsym in (2, 3, 5, 6, 8, 9, 12, 15, 16):
        vals = np.sort(rng.choice(np.arange(1, 256), nsym - 1, replace=False)).astype(np.uint8) if nsym > 1 else None
        alph = np.concatenate([[0], vals]).astype(np.uint8)
        for n in (1023, 1024, 1028, 4099, 300001):
            datas.append(alph[rng.integers(0, nsym, n)].tobytes())                      # uniform
            pr = np.full(nsym, 0.002)
            pr[0] = 1.0 - 0.002 * (nsym - 1)
            datas.append(alph[rng.choice(nsym, n, p=pr)].tobytes())                     # one symbol, rare others
            datas.append(alph[np.minimum(rng.geometric(0.45, n) - 1, nsym - 1)].tobytes())  # skewed
    datas.append(np.array([33 + (i // 3) % 7 for i in range(50000)], np.uint8).tobytes())  # no byte 0 in the data
    streams = [orc.rans(d, 1) for d in datas]
    assert engine.uncompress_many(streams) == datas
    outs = engine.compress_many(datas, [1] * len(datas))
    assert outs == streams
    # damaged payloads: the decoder must come back (an error code or garbage of the right length), never fault
    bad = []
    for s in streams[12:40:3]:
        b = bytearray(s)
        for k in range(len(b) // 2, len(b) // 2 + 40):
            b[k] ^= 0xA5
        bad.append(bytes(b))
        cut = bytes(s[:len(s) * 2 // 3])
        bad.append(cut[:1] + (len(cut) - 9).to_bytes(4, "little") + cut[5:])
    for t in bad:
        try:
            out = engine.uncompress_many([t])
            assert len(out[0]) == int.from_bytes(t[5:9], "little")
        except RuntimeError:
            pass


def test_large_quality_block(engine, orc):
    from cramtools_amd import synth
    q = synth.markov_quals(1500000, 2).tobytes()
    outs = engine.compress_many([q, q], [0, 1])
    assert outs[0] == orc.rans(q, 0) and outs[1] == orc.rans(q, 1)
    assert engine.uncompress_many(outs) == [q, q]


def test_alphabet_sample_of_long_blocks(engine, orc):
    """Order-1 blocks beyond 64 KiB take their alphabet from the first 64 KiB: a symbol that only turns up later must
    send the block through the full count (Frequencies#calcFrequencies_o1 L149-193 sees every byte)."""
    rng = np.random.default_rng(11)
    datas = []
    for wide in (False, True):
        for late in (0, 1, 2):
            d = rng.integers(0, 200, 300_000, dtype=np.uint8) if wide else rng.integers(30, 39, 300_000, dtype=np.uint8)
            if late >= 1:
                d[299_999] = 250   # the very last byte: a new symbol and a new context for nothing
            if late == 2:
                d[65_536] = 7      # the first byte behind the sample
                d[150_000:150_004] = [1, 2, 1, 2]
            datas.append(d.tobytes())
    outs = engine.compress_many(datas, [1] * len(datas))
    for d, o in zip(datas, outs):
        assert o == orc.rans(d, 1)
    assert engine.uncompress_many(outs) == datas


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
    the synchronous batch gives; one more job than CRAMGPU_MAX_JOBS (16) is refused; decode jobs round-trip."""
    rng = np.random.default_rng(3)
    batches = []
    for k in range(16):
        datas = [rng.integers(0, 5 + 15 * k, n, dtype=np.uint8).tobytes() for n in (3000 + k, 777, 5, 1200)]
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


def test_two_streams_one_engine(engine, orc):
    """Consecutive *_dev calls on DIFFERENT streams of one context share its scratch (descriptors, LPT permutation,
    per-workgroup tables): the library orders them on the device with an event, so both batches come out right even
    though nothing synchronises the two streams."""
    import torch
    from cramtools_amd import external_compression as xc
    rng = np.random.default_rng(17)
    batches = []
    for k in range(2):
        # wide alphabets use the per-workgroup global scratch, the thing two overlapping launches would trample
        datas = [rng.integers(0, 256, 300000 + 1000 * k, dtype=np.uint8).tobytes() for _ in range(6)]
        datas += [rng.integers(0, 7 + k, 50000, dtype=np.uint8).tobytes() for _ in range(6)]
        batches.append(datas)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    res = []
    for k in range(2):
        datas = batches[k]
        n = len(datas)
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode([len(d) for d in datas], [1] * n)
        h_in = np.zeros(in_bytes, np.uint8)
        for i, d in enumerate(datas):
            h_in[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(d, np.uint8)
        d_in = torch.as_tensor(h_in).to(DEV)
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=DEV)
        d_len = torch.zeros(n, dtype=torch.int32, device=DEV)
        d_pos = torch.zeros(n, dtype=torch.int64, device=DEV)
        res.append((blocks, d_in, d_out, d_len, d_pos))
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(2):
            blocks, d_in, d_out, d_len, d_pos = res[k]
            engine.encode_dev(d_in, d_out, blocks, len(batches[k]), d_len, d_pos, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        blocks, d_in, d_out, d_len, d_pos = res[k]
        lens, pos, out = d_len.cpu().numpy(), d_pos.cpu().numpy(), d_out.cpu().numpy()
        for i, d in enumerate(batches[k]):
            assert out[pos[i]:pos[i] + lens[i]].tobytes() == orc.rans(d, 1), (k, i)


def test_gather_reports_overflow(engine, orc):
    """cramgpu_gather_dev with a destination one byte too small copies nothing and says so in d_dst_off[n]."""
    import torch
    from cramtools_amd import external_compression as xc
    datas = [bytes([65 + (i * 7 + k) % 5 for i in range(3000 + 17 * k)]) for k in range(5)]
    blocks, in_bytes, out_bytes = xc.Engine.plan_encode([len(d) for d in datas], [0] * 5)
    h_in = np.zeros(in_bytes, np.uint8)
    for i, d in enumerate(datas):
        h_in[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(d, np.uint8)
    d_in = torch.as_tensor(h_in).to(DEV)
    d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=DEV)
    d_len = torch.zeros(5, dtype=torch.int32, device=DEV)
    d_pos = torch.zeros(5, dtype=torch.int64, device=DEV)
    engine.encode_dev(d_in, d_out, blocks, 5, d_len, d_pos)
    want = [orc.rans(d, 0) for d in datas]
    need = sum((len(w) + 15) // 16 * 16 for w in want)  # every piece is rounded up, the last one too
    d_off = torch.zeros(6, dtype=torch.int64, device=DEV)
    small = torch.full((need - 1,), 0x55, dtype=torch.uint8, device=DEV)
    engine.gather_dev(d_out, d_pos, d_len, 5, max(b.out_cap for b in blocks[:5]), small, d_off)
    torch.cuda.synchronize()
    assert int(d_off[5]) == need and bool((small == 0x55).all())
    exact = torch.zeros(need, dtype=torch.uint8, device=DEV)
    engine.gather_dev(d_out, d_pos, d_len, 5, max(b.out_cap for b in blocks[:5]), exact, d_off)
    torch.cuda.synchronize()
    for i, w in enumerate(want):
        assert exact[int(d_off[i]):int(d_off[i]) + len(w)].cpu().numpy().tobytes() == w
