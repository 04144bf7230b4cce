#!/usr/bin/env python
"""cramgpu_gunzip_dev at BAM scale: BGZF blocks of 65 280 bytes of BAM-like records (names, 4-bit bases, phred bytes),
zlib level 5, replicated to 20 000 blocks (1.3 GB inflated); zlib on one host core beside it."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cramtools_amd import _lib, external_compression as xc

def bgzf(data, level=5):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    return (struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1) + body +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

rng = np.random.default_rng(1)
distinct, reps = 500, 40
blocks, raws = [], []
for k in range(distinct):
    recs = []
    while sum(map(len, recs)) < 65280:
        seq = rng.integers(0, 256, 75).astype(np.uint8).tobytes()            # packed bases: near-random nybbles
        qual = np.clip(rng.normal(36, 4, 150), 2, 41).astype(np.uint8).tobytes()
        recs.append(struct.pack("<iiiBBHHHiiii", 280, 0, int(rng.integers(0, 1 << 26)), 12, 60, 4681, 1, 99, 150, 0, 1000, 300) +
                    b"read%07d\0" % int(rng.integers(0, 10 ** 7)) + struct.pack("<I", 150 << 4) + seq + qual)
    raw = b"".join(recs)[:65280]
    raws.append(raw)
    blocks.append(bgzf(raw))
n = distinct * reps
desc = (_lib.Block * n)()
io_ = oo = 0
for i in range(n):
    b = blocks[i % distinct]
    desc[i] = _lib.Block(io_, oo, len(b), 65280, 0, 0)
    io_ += len(b); oo += 65280
dev = "cuda:0"
eng = xc.Engine(0)
d_in = torch.as_tensor(np.frombuffer(b"".join(blocks) * reps + bytes(16), np.uint8).copy()).to(dev)
d_out = torch.zeros(oo + 16, dtype=torch.uint8, device=dev)
d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
d_len = torch.zeros(n, dtype=torch.int32, device=dev)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.gunzip_dev(d_in, d_out, d_desc, n, d_len)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("gunzip: %d BGZF blocks, %.1f MB in -> %.1f MB out: %.2f ms = %.1f GB/s of inflated bytes" % (n, io_ / 1e6, oo / 1e6, ms, oo / ms / 1e6), flush=True)
assert bool((d_len == 65280).all().item())
out = d_out.cpu().numpy()
for i in (0, 1, distinct - 1, n - 1):
    assert out[i * 65280:(i + 1) * 65280].tobytes() == raws[i % distinct]
t0 = time.perf_counter()
for b in blocks[:200]:
    zlib.decompress(b, 31)
dt = time.perf_counter() - t0
print("zlib, one host core: %.2f GB/s of inflated bytes (200 blocks)" % (200 * 65280 / dt / 1e9))
eng.close()
