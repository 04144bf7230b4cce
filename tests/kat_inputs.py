"""Materialise the inputs named in tests/golden/rans_kat1.json."""
import numpy as np

from cramtools_amd import synth


def kat_input(spec):
    if "ascii" in spec:
        return spec["ascii"].encode()
    if "repeat" in spec:
        return spec["repeat"].encode() * spec["n"]
    if "lcg" in spec:
        n, seed, al = spec["lcg"]
        alphabet = {"qual8": synth.QUAL8, "skew": synth.SKEW, "ACGT": b"ACGT"}[al]
        return synth.lcg_bytes(n, seed, alphabet).tobytes()
    if "dom" in spec:
        return synth.dom()
    if "markov_quals" in spec:
        n, seed = spec["markov_quals"]
        return synth.markov_quals(n, seed).tobytes()
    raise KeyError(spec)


def fuzz_inputs():
    """Edge-case corpus: sizes around the 4-way / quarter boundaries x alphabets (SURVEY.md §8(c) spec validation)."""
    out = []
    sizes = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 1000, 5000]
    alphabets = [b"A", b"AC", b"ACGT", synth.QUAL8, bytes(range(256))]
    for n in sizes:
        for k, al in enumerate(alphabets):
            out.append(synth.lcg_bytes(n, 7 + n + k, al).tobytes())
    out.append(bytes(range(256)) * 3)
    out.append(b"\x00" * 777)
    out.append(b"\xff" * 5 + b"\x00" * 5)
    return out
