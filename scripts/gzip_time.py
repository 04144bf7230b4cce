#!/usr/bin/env python
"""cramgpu_gzip_dev at scale.  (1) BAM side: 20 000 blocks of 65 280 bytes of BAM-like records -> BGZF members; the image
is inflated again by the gunzip kernel (all blocks) and by zlib (a sample).  (2) CRAM side: 10 000 series-like streams of
8-40 KB (ITF8-shaped values, flags, names) -> gzip members.  Sizes and times beside zlib level 5 on one host core."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cramtools_amd import _lib, external_compression as xc

rng = np.random.default_rng(1)
dev = "cuda:0"
eng = xc.Engine(0)


def run(label, raws, reps, bgzf):
    distinct = len(raws)
    n = distinct * reps
    desc = (_lib.Block * n)()
    io_ = oo = 0
    for i in range(n):
        r = raws[i % distinct]
        cap = min(eng.gzip_bound(len(r)), 65536) if bgzf else eng.gzip_bound(len(r))
        desc[i] = _lib.Block(io_, oo, len(r), cap, 0, 0)
        io_ += len(r); oo += cap
    d_in = torch.as_tensor(np.frombuffer(b"".join(raws) * reps + bytes(16), np.uint8).copy()).to(dev)
    d_out = torch.zeros(oo + 16, dtype=torch.uint8, device=dev)
    d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
    d_len = torch.zeros(n, dtype=torch.int32, device=dev)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.gzip_dev(d_in, d_out, d_desc, n, d_len, bgzf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        comp = int(d_len.clamp(min=0).sum().item())
        print("%s: %d streams, %.1f MB -> %.1f MB: %.2f ms = %.1f GB/s of input" % (label, n, io_ / 1e6, comp / 1e6, ms, io_ / ms / 1e6), flush=True)
    lens = d_len.cpu().numpy()
    assert (lens > 0).all()
    # every member back through the gunzip kernel
    back = (_lib.Block * n)()
    for i in range(n):
        back[i] = _lib.Block(desc[i].out_off, desc[i].in_off, int(lens[i]), desc[i].in_len, 0, 0)
    d_back = torch.zeros(io_ + 16, dtype=torch.uint8, device=dev)
    d_bdesc = torch.as_tensor(np.frombuffer(bytes(back), np.uint8).copy()).to(dev)
    d_blen = torch.zeros(n, dtype=torch.int32, device=dev)
    eng.gunzip_dev(d_out, d_back, d_bdesc, n, d_blen)
    torch.cuda.synchronize()
    assert torch.equal(d_back[:io_], d_in[:io_]), "round trip through the gunzip kernel"
    out = d_out.cpu().numpy()
    t0 = time.perf_counter()
    zsize = 0
    sample = list(range(0, distinct, max(1, distinct // 200)))
    for i in sample:
        zsize += len(zlib.compress(raws[i], 5)) + 12 + (6 if bgzf else 0)
    dt = time.perf_counter() - t0
    mine = 0
    for i in sample:
        m = out[desc[i].out_off:desc[i].out_off + int(lens[i])].tobytes()
        assert zlib.decompress(m, 31) == raws[i]
        mine += len(m)
    raw = sum(len(raws[i]) for i in sample)
    print("%s: sample of %d streams (%.2f MB): kernel %.3f MB, zlib level 5 %.3f MB (ratio %.3f); zlib on one host core %.3f GB/s"
          % (label, len(sample), raw / 1e6, mine / 1e6, zsize / 1e6, mine / zsize, raw / dt / 1e9), flush=True)


bam = []
for k in range(500):
    recs = []
    while sum(map(len, recs)) < 65280:
        seq = rng.integers(0, 256, 75).astype(np.uint8).tobytes()
        qual = np.clip(rng.normal(36, 4, 150), 2, 41).astype(np.uint8).tobytes()
        recs.append(struct.pack("<iiiBBHHHiiii", 280, 0, int(rng.integers(0, 1 << 26)), 12, 60, 4681, 1, 99, 150, 0, 1000, 300) +
                    b"read%07d\0" % int(rng.integers(0, 10 ** 7)) + struct.pack("<I", 150 << 4) + seq + qual)
    bam.append(b"".join(recs)[:65280])
run("BGZF", bam, 40, True)

series = []
for k in range(500):
    kind = k % 4
    if kind == 0:   # flags-like: a few values
        s = rng.choice(np.array([99, 147, 83, 163, 0, 16], np.uint8), 10000, p=[0.3, 0.3, 0.15, 0.15, 0.05, 0.05]).tobytes()
    elif kind == 1:  # ITF8 two-byte values around an insert size
        v = np.clip(rng.normal(300, 40, 10000), 128, 16000).astype(np.int64)
        s = np.stack([0x80 | (v >> 8), v & 255], 1).astype(np.uint8).tobytes()
    elif kind == 2:  # near-random 4-byte positions
        s = rng.integers(0, 1 << 26, 10000).astype("<u4").tobytes()
    else:            # names
        s = b"".join(b"read.%d/%d\0" % (int(x), int(x) % 7) for x in rng.integers(0, 10 ** 7, 800))
    series.append(s)
run("series", series, 20, False)
eng.close()
