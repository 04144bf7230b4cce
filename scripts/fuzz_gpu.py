#!/usr/bin/env python
"""Wide fuzz of the rANS kernels against the oracle: random sizes, alphabets, run structures and ITF8-like series, both
orders; every stream byte-identical to the oracle's, every decode identical to the input."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cramtools_amd import external_compression as xc
from oracle import orc

def gen(rng, kind, n):
    if kind == 0:
        return rng.integers(0, rng.integers(1, 257), n, dtype=np.uint8)
    if kind == 1:
        return np.repeat(rng.integers(0, rng.integers(2, 60), n // 3 + 1, dtype=np.uint8), 3)[:n]
    if kind == 2:
        return (rng.geometric(rng.uniform(0.05, 0.6), n) % 256).astype(np.uint8)
    if kind == 3:  # ITF8-like: wide alphabet, sparse contexts
        v = rng.integers(120, 20000, n // 2 + 1)
        b = np.stack([0x80 | (v >> 8), v & 0xFF], 1).astype(np.uint8).reshape(-1)
        return b[:n]
    q = rng.choice(np.array([2, 11, 25, 37, 40], np.uint8), n, p=[0.05, 0.1, 0.15, 0.3, 0.4])
    return q

def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    orc.build()
    eng = xc.Engine(0)
    total = 0
    for rnd in range(6):
        datas = []
        for _ in range(400):
            n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 5000), rng.integers(5000, 200000)]))
            datas.append(gen(rng, int(rng.integers(0, 5)), n).tobytes())
        for order in (0, 1):
            outs = eng.compress_many(datas, [order] * len(datas))
            for i, (d, o) in enumerate(zip(datas, outs)):
                assert o == orc.rans(d, order), ("encode", rnd, order, i, len(d))
            back = eng.uncompress_many(outs)
            for i, (d, b) in enumerate(zip(datas, back)):
                assert d == b, ("decode", rnd, order, i, len(d))
            total += len(datas)
        print("round %d ok (%d blocks so far)" % (rnd, total), flush=True)
    eng.close()
    print("fuzz ok: %d blocks, both kernels, byte-identical to the oracle" % total)

if __name__ == "__main__":
    main()
