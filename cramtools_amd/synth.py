"""Seeded synthetic inputs: LCG byte streams, Markov quality strings, reads and the
per-container external-block corpus.  Everything derives from `lcg`-style streams so any host reproduces the
same bytes.  Vectorised with numpy (the LCG is advanced by jump-ahead doubling).
"""
import numpy as np

_A = 1103515245
_C = 12345
_M = 0x7FFFFFFF

QUAL8 = bytes([40] * 20 + [37] * 6 + [33] * 4 + [27] * 2 + [22, 15, 6, 2])
SKEW = b"A" * 10 + b"C" * 7 + b"G" * 7 + b"T" * 10 + b"N"


def dom():
    d = bytearray(4745)
    for i in range(1, 256):
        d[18 * i] = i
    return bytes(d)


def lcg_stream(n, seed):
    """x_1..x_n of x = (x*1103515245 + 12345) & 0x7FFFFFFF starting from x_0 = seed (uint64 array)."""
    n = int(n)
    out = np.empty(max(n, 1), np.uint64)
    out[0] = (int(seed) * _A + _C) & _M
    have = 1
    a_k, c_k = _A & _M, _C  # x_{j+1} = a_k x_j + c_k
    while have < n:
        take = min(have, n - have)
        out[have:have + take] = (out[:take] * np.uint64(a_k) + np.uint64(c_k)) & np.uint64(_M)
        have += take
        c_k = (a_k * c_k + c_k) & _M
        a_k = (a_k * a_k) & _M
    return out[:n]


def lcg_bytes(n, seed, alphabet):
    al = np.frombuffer(bytes(alphabet), np.uint8)
    x = lcg_stream(n, seed)
    return al[((x >> np.uint64(16)) % np.uint64(al.size)).astype(np.int64)]


def markov_quals(n, seed):
    """lut=[2,11,25,37,37,40,40,40]; q=40; per step: x=lcg(x); if ((x>>8)&3)==0: q=lut[(x>>16)%8]; emit q."""
    lut = np.array([2, 11, 25, 37, 37, 40, 40, 40], np.uint8)
    x = lcg_stream(n, seed)
    ev = ((x >> np.uint64(8)) & np.uint64(3)) == 0
    val = lut[((x >> np.uint64(16)) % np.uint64(8)).astype(np.int64)]
    idx = np.where(ev, np.arange(x.size, dtype=np.int64), -1)
    last = np.maximum.accumulate(idx)
    out = np.where(last >= 0, val[np.maximum(last, 0)], np.uint8(40)).astype(np.uint8)
    return out


def itf8(values):
    """ITF8#writeUnsignedITF8 (htsjdk!io/ITF8 L53-84) of an int array -> uint8 array."""
    v = np.asarray(values, np.int64) & 0xFFFFFFFF
    n1 = v < (1 << 7)
    n2 = (~n1) & (v < (1 << 14))
    n3 = (~n1) & (~n2) & (v < (1 << 21))
    n4 = (~n1) & (~n2) & (~n3) & (v < (1 << 28))
    n5 = ~(n1 | n2 | n3 | n4)
    ln = (n1 * 1 + n2 * 2 + n3 * 3 + n4 * 4 + n5 * 5).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
    out = np.zeros(int(off[-1]), np.uint8)
    o = off[:-1]

    def put(mask, k, vals):
        out[o[mask] + k] = (vals & 0xFF).astype(np.uint8)

    put(n1, 0, v[n1])
    put(n2, 0, (v[n2] >> 8) | 0x80)
    put(n2, 1, v[n2])
    put(n3, 0, (v[n3] >> 16) | 0xC0)
    put(n3, 1, v[n3] >> 8)
    put(n3, 2, v[n3])
    put(n4, 0, (v[n4] >> 24) | 0xE0)
    put(n4, 1, v[n4] >> 16)
    put(n4, 2, v[n4] >> 8)
    put(n4, 3, v[n4])
    put(n5, 0, (v[n5] >> 28) | 0xF0)
    put(n5, 1, v[n5] >> 20)
    put(n5, 2, v[n5] >> 12)
    put(n5, 3, v[n5] >> 4)
    put(n5, 4, v[n5] & 0x0F)
    return out


def reference(length, seed=0xC0FFEE):
    return lcg_bytes(length, seed, b"ACGT")


class ReadSet:
    """Flat, start-sorted synthetic reads in the SoA layout of include/cramgpu.h."""
    pass


def make_reads(n_reads, ref, read_len=150, seed=1, sub_rate=0.001, indel_rate=0.0001, softclip_frac=0.01,
               ins_rate=None, del_rate=None, qual_mode="markov", qual_seed=2, err_seed=3):
    """Illumina-like (default) or long-read (high indel) reads against `ref` (uint8 array).

    Reads are built from simple per-read edit scripts drawn from lcg streams: a read is a walk over the reference
    with substitutions (per aligned base), insertions of 1-3 bp and deletions of 1-5 bp (per aligned base, in the
    read interior), and an optional leading soft clip of 1-20 bp.  Returns a ReadSet.
    """
    ref = np.asarray(ref, np.uint8)
    L = int(read_len)
    n = int(n_reads)
    ins_rate = indel_rate / 2 if ins_rate is None else ins_rate
    del_rate = indel_rate / 2 if del_rate is None else del_rate
    rs = lcg_stream(n, seed)
    span_max = L + 64 if (ins_rate + del_rate) < 0.01 else int(L * 1.3) + 64
    starts = np.sort((rs >> np.uint64(4)) % np.uint64(max(ref.size - span_max, 1))).astype(np.int64) + 1
    flags_rnd = lcg_stream(n, seed + 1000)
    clip_on = ((flags_rnd >> np.uint64(8)) % np.uint64(10000)) < int(softclip_frac * 10000)
    clip_len = np.where(clip_on, 1 + ((flags_rnd >> np.uint64(20)) % np.uint64(20)), 0).astype(np.int64)
    mapq = np.where(((flags_rnd >> np.uint64(3)) % np.uint64(100)) < 5, 0, 60).astype(np.uint8)

    # per aligned-base event draws: one stream, cut per read (L draws per read is an upper bound)
    ev = lcg_stream(n * L, err_seed).reshape(n, L)
    u = ((ev >> np.uint64(7)) % np.uint64(1000000)).astype(np.int64)
    is_sub = u < int(sub_rate * 1e6)
    is_ins = (~is_sub) & (u < int((sub_rate + ins_rate) * 1e6))
    is_del = (~is_sub) & (~is_ins) & (u < int((sub_rate + ins_rate + del_rate) * 1e6))
    aux = ((ev >> np.uint64(3)) % np.uint64(15)).astype(np.int64)
    alpha = np.frombuffer(b"ACGT", np.uint8)

    bases = np.empty(n * L, np.uint8)
    cig_ops = []
    cig_off = np.zeros(n + 1, np.int64)
    simple = ~(is_ins.any(axis=1) | is_del.any(axis=1) | clip_on)
    # fast path: reads with substitutions only
    idx = starts[:, None] - 1 + np.arange(L)[None, :]
    rb = ref[np.minimum(idx, ref.size - 1)]
    subbed = alpha[(np.searchsorted(alpha, rb) + 1 + aux % 3) % 4]
    rb = np.where(is_sub, subbed, rb)
    bases[:] = rb.reshape(-1)
    cigars = [None] * n
    for r in range(n):
        if simple[r]:
            cigars[r] = [(L << 4) | 0]
            continue
        # slow path: explicit edit script
        out = []
        ops = []
        rp = int(starts[r]) - 1
        cl = int(clip_len[r])

        def add(op, ln):
            if ln <= 0:
                return
            if ops and ops[-1][0] == op:
                ops[-1][1] += ln
            else:
                ops.append([op, ln])

        for k in range(cl):
            out.append(alpha[int(aux[r, k % L]) % 4])
        add(4, cl)
        k = 0
        while len(out) < L:
            e = k % L
            k += 1
            interior = cl + 2 < len(out) < L - 4 and ops and ops[-1][0] == 0
            if is_ins[r, e] and interior:
                ln = min(1 + int(aux[r, e]) % 3, L - len(out) - 1)
                for t in range(ln):
                    out.append(alpha[(int(aux[r, e]) + t) % 4])
                add(1, ln)
                continue
            if is_del[r, e] and interior:
                ln = 1 + int(aux[r, e]) % 5
                rp += ln
                add(2, ln)
                # a deletion must be followed by an aligned base
                b = ref[min(rp, ref.size - 1)]
                out.append(b)
                rp += 1
                add(0, 1)
                continue
            b = ref[min(rp, ref.size - 1)]
            if is_sub[r, e]:
                b = alpha[(int(np.searchsorted(alpha, b)) + 1 + int(aux[r, e]) % 3) % 4]
            out.append(b)
            rp += 1
            add(0, 1)
        bases[r * L:(r + 1) * L] = np.asarray(out[:L], np.uint8)
        cigars[r] = [(ln << 4) | op for op, ln in ops]
    nc = np.array([len(c) for c in cigars], np.int64)
    cig_off[1:] = np.cumsum(nc)
    cigar = np.array([x for c in cigars for x in c], np.uint32)

    out = ReadSet()
    out.n = n
    out.read_len = np.full(n, L, np.int32)
    out.base_off = (np.arange(n, dtype=np.int64) * L)
    out.bases = bases
    out.aln_start = starts.astype(np.int32)
    out.cigar = cigar
    out.cigar_off = cig_off[:-1].copy()
    out.n_cigar = nc.astype(np.int32)
    out.mapq = mapq
    out.rflags = np.zeros(n, np.uint8)
    if qual_mode == "markov":
        out.quals = markov_quals(n * L, qual_seed)
    else:  # uniform 1..30 (config C5)
        out.quals = (1 + (lcg_stream(n * L, qual_seed) >> np.uint64(16)) % np.uint64(30)).astype(np.uint8)
    ref_len = np.zeros(n, np.int64)
    for r in range(n):
        ref_len[r] = sum(int(c) >> 4 for c in cigars[r] if (int(c) & 15) in (0, 2, 3, 7, 8))
    out.aln_span = (ref_len - 1).astype(np.int32)
    return out
