#!/usr/bin/env python
"""Slice reference MD5 at the configs[1] shape: 1000 slices of 64 000 reference bases each (cramgpu_slice_md5_dev)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cramtools_amd import bam, external_compression as xc

dev = "cuda:0"
eng = xc.Engine(0)
n, span = 1000, 64000
ref = torch.randint(65, 85, (n * span + 16,), dtype=torch.uint8, device=dev)
start = (np.arange(n, dtype=np.int64) * span + 1).astype(np.int32)
wins = [(0, n * span, 0)] * n
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = bam.slice_ref_md5(eng, dev, ref, wins, start, np.full(n, span, np.int32))
    e1.record()
    torch.cuda.synchronize()
    print("slice MD5: %d slices x %d bases: %.3f ms (includes the wrapper's small uploads)" % (n, span, e0.elapsed_time(e1)), flush=True)
h = ref.cpu().numpy()
got = out.cpu().numpy()
for s in (0, 1, 499, 999):
    assert got[s].tobytes() == hashlib.md5(h[s * span:(s + 1) * span].tobytes()).digest()
print("checked against hashlib")
eng.close()
