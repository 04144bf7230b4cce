"""Test helper: a small pure-Python rANS 4x8 ENCODER with caller-chosen frequency tables, written from SURVEY.md §8(a)
rows R4-R8.  It makes valid CRAM 3.0 streams that htsjdk's own encoder never emits but its decoder accepts (tables that
sum to 4096 as htslib writes them, slot 4095 in use, a frequency of 4096 in both of its spellings), so that the decoders
are exercised beyond the encoder's habits.  The oracle decodes every stream first; only then is it shown to the GPU."""
import numpy as np

L = 1 << 23


def normalise_to(counts, total=4096):
    """counts[256] -> frequencies summing to exactly `total` (every present symbol >= 1)."""
    c = np.asarray(counts, np.int64)
    f = np.where(c > 0, np.maximum(1, c * total // max(int(c.sum()), 1)), 0)
    f[int(np.argmax(c))] += total - int(f.sum())
    assert f.sum() == total and (f[c > 0] > 0).all()
    return f


def _put(state, out, start, freq):
    x_max = ((L >> 12) << 8) * freq
    while state >= x_max:
        out.append(state & 0xFF)
        state >>= 8
    return ((state // freq) << 12) + (state % freq) + start


def _row(F, spell_4096_as_zero):
    b, rle = bytearray(), 0
    for j in range(256):
        if not F[j]:
            continue
        if rle:
            rle -= 1
        else:
            b.append(j)
            if j > 0 and F[j - 1]:
                k = j + 1
                while k < 256 and F[k]:
                    k += 1
                rle = k - (j + 1)
                b.append(rle)
        f = int(F[j])
        if f == 4096 and spell_4096_as_zero:
            b.append(0)
        elif f < 128:
            b.append(f)
        else:
            b += bytes([0x80 | (f >> 8), f & 0xFF])
    b.append(0)
    return bytes(b)


def _finish(order, n, table, out, states):
    for s in (states[3], states[2], states[1], states[0]):
        out += s.to_bytes(4, "big")
    payload = bytes(out)[::-1]
    body = table + payload
    return bytes([order]) + len(body).to_bytes(4, "little") + n.to_bytes(4, "little") + body


def encode_o0(data, F):
    d = bytes(data)
    n = len(d)
    C = np.concatenate([[0], np.cumsum(F)[:-1]])
    r = [L, L, L, L]
    out = bytearray()
    t = n & 3
    for k in range(t):  # tail: rans(t-1) <- in[n-1], ..., rans0 <- in[n-t]
        s = d[n - 1 - k]
        r[t - 1 - k] = _put(r[t - 1 - k], out, int(C[s]), int(F[s]))
    i = n & ~3
    while i > 0:
        for k in range(4):
            s = d[i - 1 - k]
            r[3 - k] = _put(r[3 - k], out, int(C[s]), int(F[s]))
        i -= 4
    return _finish(0, n, _row(F, False), out, r)


def encode_o1(data, F2, spell_4096_as_zero=False):
    """F2[ctx][sym]; rows of used contexts sum to 4096."""
    d = bytes(data)
    n = len(d)
    C2 = np.concatenate([np.zeros((256, 1), np.int64), np.cumsum(F2, 1)[:, :-1]], 1)
    q = n >> 2
    i = [q - 2, 2 * q - 2, 3 * q - 2, n - 2]
    l = [d[i[0] + 1] if i[0] + 1 >= 0 else 0, d[i[1] + 1] if i[1] + 1 >= 0 else 0, d[i[2] + 1] if i[2] + 1 >= 0 else 0, d[n - 1]]
    r = [L, L, L, L]
    out = bytearray()

    def put(k, ctx, sym):
        assert F2[ctx][sym] > 0, (ctx, sym)
        r[k] = _put(r[k], out, int(C2[ctx][sym]), int(F2[ctx][sym]))

    while i[3] > 4 * q - 2 and i[3] >= 0:
        c = d[i[3]]
        put(3, c, l[3])
        l[3] = c
        i[3] -= 1
    while i[0] >= 0:
        for k in (3, 2, 1, 0):
            c = d[i[k]]
            put(k, c, l[k])
            l[k] = c
            i[k] -= 1
    for k in (3, 2, 1, 0):
        put(k, 0, l[k])
    T = F2.sum(1)
    tb, rle = bytearray(), 0
    for c in range(256):
        if not T[c]:
            continue
        if rle:
            rle -= 1
        else:
            tb.append(c)
            if c > 0 and T[c - 1]:
                k = c + 1
                while k < 256 and T[k]:
                    k += 1
                rle = k - (c + 1)
                tb.append(rle)
        tb += _row(F2[c], spell_4096_as_zero)
    tb.append(0)
    return _finish(1, n, bytes(tb), out, r)


def o1_tables(data, total=4096):
    """Order-1 counts as Frequencies#calcFrequencies_o1 L149-193 takes them, normalised to `total` per context."""
    d = bytes(data)
    n = len(d)
    cnt = np.zeros((256, 256), np.int64)
    last = 0
    for c in d:
        cnt[last][c] += 1
        last = c
    q = n >> 2
    for k in (1, 2, 3):
        cnt[0][d[k * q]] += 1
    F2 = np.zeros((256, 256), np.int64)
    for c in range(256):
        if cnt[c].sum():
            F2[c] = normalise_to(cnt[c], total)
    return F2


def cases():
    """[(name, data, stream)]"""
    rng = np.random.default_rng(17)
    out = []
    a = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 1003, p=[0.7, 0.1, 0.1, 0.1]))
    out.append(("o0 sum 4096, n%4==3", a, encode_o0(a, normalise_to(np.bincount(np.frombuffer(a, np.uint8), minlength=256)))))
    b = bytes(rng.integers(0, 200, 2050, dtype=np.uint8))
    out.append(("o0 wide sum 4096, n%4==2", b, encode_o0(b, normalise_to(np.bincount(np.frombuffer(b, np.uint8), minlength=256)))))
    out.append(("o1 sum 4096", a, encode_o1(a, o1_tables(a))))
    m = bytes(np.repeat(rng.integers(30, 42, 400, dtype=np.uint8), 5))[:1999]
    out.append(("o1 runs sum 4096, n%4==3", m, encode_o1(m, o1_tables(m))))
    c = bytes(rng.integers(0, 90, 3001, dtype=np.uint8))
    out.append(("o1 90 symbols sum 4096", c, encode_o1(c, o1_tables(c))))
    one = b"A" * 50
    out.append(("o1 single symbol F=4096 (two bytes)", one, encode_o1(one, o1_tables(one), False)))
    out.append(("o1 single symbol F=4096 (spelled 0)", one, encode_o1(one, o1_tables(one), True)))
    return out
