"""cramgpu_rans_encode_static — the encode entry point that allocates nothing (static device scratch) — against
cramgpu_rans_encode_dev on the same inputs: byte-identical streams, lengths and positions, on two calls in a row, the
second with a smaller input (the shared static scratch keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _batch(rng, sizes):
    datas, orders = [], []
    for k, n in enumerate(sizes):
        kind = k % 4
        if kind == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8)          # wide alphabet: per-workgroup tables in the static scratch
        elif kind == 1:
            d = rng.integers(0, 6, n, dtype=np.uint8) * 7 + 2    # quality-like
        elif kind == 2:
            d = (rng.geometric(0.2, n) % 200).astype(np.uint8)   # ITF8-like
        else:
            d = np.repeat(rng.integers(0, 40, n // 9 + 1, dtype=np.uint8), 9)[:n]
        datas.append(d.tobytes())
        orders.append(0 if k % 5 == 4 else 1)
    return datas, orders


def _run_both(eng, xc, _lib, torch, datas, orders):
    n = len(datas)
    blocks, in_bytes, out_bytes = xc.Engine.plan_encode([len(d) for d in datas], orders)
    h_in = np.zeros(in_bytes, np.uint8)
    for i, d in enumerate(datas):
        h_in[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(d, np.uint8)
    d_in = torch.as_tensor(h_in).to(DEV)
    res = []
    d_blocks = torch.as_tensor(np.frombuffer(bytes(blocks), np.uint8)[:n * C.sizeof(_lib.Block)].copy()).to(DEV)
    for static in (False, True):
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=DEV)
        d_len = torch.full((n,), -77, dtype=torch.int32, device=DEV)
        d_pos = torch.zeros(n, dtype=torch.int64, device=DEV)
        s = torch.cuda.current_stream().cuda_stream
        if static:
            _lib.check(_lib.lib().cramgpu_rans_encode_static(eng.handle, d_in.data_ptr(), d_out.data_ptr(), d_blocks.data_ptr(),
                                                             n, d_len.data_ptr(), d_pos.data_ptr(), s))
        else:
            eng.encode_dev(d_in, d_out, blocks, n, d_len, d_pos)
        torch.cuda.synchronize()
        res.append((d_out.cpu().numpy(), d_len.cpu().numpy(), d_pos.cpu().numpy()))
    (o0, l0, p0), (o1, l1, p1) = res
    assert (l0 > 0).all()
    assert np.array_equal(l0, l1) and np.array_equal(p0, p1)
    for i in range(n):
        assert np.array_equal(o0[p0[i]:p0[i] + l0[i]], o1[p1[i]:p1[i] + l1[i]]), i
    return [o1[p1[i]:p1[i] + l1[i]].tobytes() for i in range(n)]


def test_static_scratch_entry_point_matches_encode_dev(orc):
    import torch
    from cramtools_amd import _lib
    from cramtools_amd import external_compression as xc
    eng = xc.Engine(0)
    try:
        rng = np.random.default_rng(77)
        # first call: 90 blocks (more than the 32 workgroups of the static path), several wide-alphabet ones
        datas, orders = _batch(rng, [int(x) for x in rng.integers(1, 120000, 90)] + [400000, 3, 0 + 1])
        got = _run_both(eng, xc, _lib, torch, datas, orders)
        for d, o, g in zip(datas[:12], orders[:12], got[:12]):
            assert g == orc.rans(d, o)
        # second call, smaller: the scratch still holds the first call's permutation and tables
        datas, orders = _batch(rng, [70000, 150, 9000, 31000, 5])
        got = _run_both(eng, xc, _lib, torch, datas, orders)
        for d, o, g in zip(datas, orders, got):
            assert g == orc.rans(d, o)
        # descriptors come from device memory, so the host cannot check them: the kernel does, block by block — a misaligned
        # offset, a negative length or capacity give that block CRAMGPU_E_ARG and leave its neighbours alone
        datas, orders = _batch(rng, [5000, 6000, 7000, 8000, 900])
        n = len(datas)
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode([len(d) for d in datas], orders)
        h_in = np.zeros(in_bytes, np.uint8)
        for i, d in enumerate(datas):
            h_in[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(d, np.uint8)
        blocks[0].in_off += 8
        blocks[1].in_len = -5
        blocks[2].out_cap = -1
        blocks[3].out_off += 4
        d_in = torch.as_tensor(h_in).to(DEV)
        d_blocks = torch.as_tensor(np.frombuffer(bytes(blocks), np.uint8)[:n * C.sizeof(_lib.Block)].copy()).to(DEV)
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=DEV)
        d_len = torch.full((n,), -77, dtype=torch.int32, device=DEV)
        d_pos = torch.zeros(n, dtype=torch.int64, device=DEV)
        _lib.check(_lib.lib().cramgpu_rans_encode_static(eng.handle, d_in.data_ptr(), d_out.data_ptr(), d_blocks.data_ptr(), n,
                                                         d_len.data_ptr(), d_pos.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        lens, pos = d_len.cpu().numpy(), d_pos.cpu().numpy()
        assert lens[0] == _lib.E_ARG and lens[1] == _lib.E_ARG and lens[3] == _lib.E_ARG and lens[2] < 0, lens
        assert d_out[int(pos[4]):int(pos[4]) + int(lens[4])].cpu().numpy().tobytes() == orc.rans(datas[4], orders[4])
        # beyond the maximum: a clear error, nothing launched
        d = torch.zeros(64, dtype=torch.uint8, device=DEV)
        rc = _lib.lib().cramgpu_rans_encode_static(eng.handle, d.data_ptr(), d.data_ptr(), d.data_ptr(), 4097, d.data_ptr(),
                                                   d.data_ptr(), None)
        assert rc == _lib.E_ARG and "4096" in _lib.last_error()
    finally:
        eng.close()
