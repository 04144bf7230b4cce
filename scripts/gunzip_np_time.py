#!/usr/bin/env python
"""The long pole of the bench's gunzip leg in isolation: a 40 KB series of 4-byte ITF8 mate positions (sorted starts + small
offsets), zlib level 5, 1000 streams (one per container)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cramtools_amd import _lib, external_compression as xc

rng = np.random.default_rng(1)
def itf8_4(v):
    return bytes([0xE0 | (v >> 24), (v >> 16) & 255, (v >> 8) & 255, v & 255])
streams, raws = [], []
for c in range(50):
    st = np.sort(rng.integers(1 << 21, 1 << 26, 10000)) + rng.integers(100, 500, 10000)
    raw = b"".join(itf8_4(int(v)) for v in st)
    raws.append(raw)
    co = zlib.compressobj(5, zlib.DEFLATED, 31)
    streams.append(co.compress(raw) + co.flush())
n = 1000
desc = (_lib.Block * n)()
io_ = oo = 0
for i in range(n):
    z = streams[i % 50]
    desc[i] = _lib.Block(io_, oo, len(z), 40000, 0, 0)
    io_ += len(z); oo += 40000
dev = "cuda:0"
eng = xc.Engine(0)
d_in = torch.as_tensor(np.frombuffer(b"".join(streams) * 20 + bytes(16), np.uint8).copy()).to(dev)
d_out = torch.zeros(oo + 16, dtype=torch.uint8, device=dev)
d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
d_len = torch.zeros(n, dtype=torch.int32, device=dev)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.gunzip_dev(d_in, d_out, d_desc, n, d_len)
    e1.record()
    torch.cuda.synchronize()
print("gunzip NP-like: %d streams, %d -> %d bytes each: %.2f ms (lens ok: %s)" % (n, len(streams[0]), 40000, e0.elapsed_time(e1), bool((d_len == 40000).all().item())), flush=True)
eng.close()
