"""Random slices for the data-series writer / reader tests (oracle and GPU): every operator of SURVEY.md Appendix A,
detached / mate-downstream / plain records, unmapped records with and without bases, values of every ITF8 width."""
import numpy as np

OPS = b"BXISHPDNiQ"


def random_slice(rng, n, max_len=60, feature_rate=0.6, unmapped_rate=0.1):
    read_len = rng.integers(1, max_len + 1, n).astype(np.int32)
    base_off = np.concatenate([[0], np.cumsum(read_len)]).astype(np.int64)
    nb = int(base_off[-1])
    bases = rng.choice(np.frombuffer(b"ACGTN", np.uint8), nb + 16)
    quals = rng.integers(0, 60, nb + 16).astype(np.uint8)
    aln_start = np.sort(rng.integers(1, 1 << 22, n)).astype(np.int32)
    unm = rng.random(n) < unmapped_rate
    bam_flags = (rng.integers(0, 1 << 11, n) & ~4).astype(np.int32) | np.where(unm, 4, 0).astype(np.int32)
    kind = rng.integers(0, 3, n)  # 0 plain, 1 detached, 2 mate downstream
    comp_flags = (np.where(kind == 1, 2, 0) | np.where(kind == 2, 4, 0) | np.where(rng.random(n) < 0.7, 1, 0) |
                  np.where(unm & (rng.random(n) < 0.3), 8, 0)).astype(np.int32)
    wide = lambda: np.where(rng.random(n) < 0.2, rng.integers(-5, 1 << 30, n), rng.integers(0, 300, n)).astype(np.int32)
    cols = dict(bam_flags=bam_flags, comp_flags=comp_flags, ref_id=rng.integers(-1, 25, n).astype(np.int32),
                read_group=rng.integers(-1, 4, n).astype(np.int32), mate_flags=rng.integers(0, 4, n).astype(np.int32),
                mate_ref=rng.integers(-1, 25, n).astype(np.int32), mate_start=wide(), tlen=(wide() - 150).astype(np.int32),
                next_frag=rng.integers(0, 2000, n).astype(np.int32), tag_list=rng.integers(0, 200, n).astype(np.int32))
    mapq = rng.integers(0, 256, n).astype(np.uint8)
    off, op, pos, a, b = [0], [], [], [], []
    for r in range(n):
        L = int(read_len[r])
        k = 0
        if not unm[r] and rng.random() < feature_rate:
            k = int(rng.integers(1, min(L, 8) + 1))
            ps = np.sort(rng.integers(1, L + 1, k))
            for p in ps:
                o = OPS[int(rng.integers(0, len(OPS)))]
                op.append(o)
                pos.append(int(p))
                if o == ord("X"):
                    a.append(int(rng.integers(65, 85)) | (int(rng.integers(65, 85)) << 8) | (int(rng.integers(0, 4)) << 16))
                    b.append(0)
                elif o in b"IS":
                    ln = int(rng.integers(1, L - p + 2))
                    a.append(int(p) - 1)
                    b.append(ln)
                elif o in b"HPDN":
                    a.append(int(rng.integers(1, 1 << int(rng.integers(1, 30)))))
                    b.append(0)
                elif o == ord("B"):
                    a.append(int(rng.integers(65, 85)))
                    b.append(int(rng.integers(0, 60)))
                else:  # i, Q
                    a.append(int(rng.integers(0, 90)))
                    b.append(0)
        off.append(off[-1] + k)
    feats = dict(off=np.array(off, np.int64), op=np.array(op + [0], np.uint8)[:len(op)] if op else np.zeros(0, np.uint8),
                 pos=np.array(pos, np.int32), a=np.array(a, np.int32), b=np.array(b, np.int32))
    return dict(n=n, read_len=read_len, base_off=base_off, bases=bases, quals=quals, aln_start=aln_start, cols=cols, mapq=mapq,
                feats=feats)


def py_itf8(v):
    """<2^7 1 B; <2^14 2 B 0x80|hi; <2^21 3 B 0xC0|; <2^28 4 B 0xE0|; else 0xF0|v>>28, v>>20, v>>12, v>>4, v&0x0F."""
    v &= 0xFFFFFFFF
    if v < 1 << 7:
        return bytes([v])
    if v < 1 << 14:
        return bytes([0x80 | (v >> 8), v & 255])
    if v < 1 << 21:
        return bytes([0xC0 | (v >> 16), (v >> 8) & 255, v & 255])
    if v < 1 << 28:
        return bytes([0xE0 | (v >> 24), (v >> 16) & 255, (v >> 8) & 255, v & 255])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 255, (v >> 12) & 255, (v >> 4) & 255, v & 15])


def py_write_series(s, multiref, ap_base, ap_delta, lo=0, hi=None):
    """A second, independent statement of Appendix A's field order in plain Python (small slices only)."""
    from oracle.orc import SERIES_ID as ID
    hi = s["n"] if hi is None else hi
    out = [bytearray() for _ in range(30)]
    c, f = s["cols"], s["feats"]
    prev = ap_base
    for r in range(lo, hi):
        cf, L = int(c["comp_flags"][r]), int(s["read_len"][r])
        out[ID["BF"]] += py_itf8(int(c["bam_flags"][r]))
        out[ID["CF"]] += py_itf8(cf)
        if multiref:
            out[ID["RI"]] += py_itf8(int(c["ref_id"][r]))
        out[ID["RL"]] += py_itf8(L)
        st = int(s["aln_start"][r])
        out[ID["AP"]] += py_itf8(st - prev if ap_delta else st)
        prev = st
        out[ID["RG"]] += py_itf8(int(c["read_group"][r]))
        if cf & 2:
            for k, col in (("MF", "mate_flags"), ("NS", "mate_ref"), ("NP", "mate_start"), ("TS", "tlen")):
                out[ID[k]] += py_itf8(int(c[col][r]))
        elif cf & 4:
            out[ID["NF"]] += py_itf8(int(c["next_frag"][r]))
        out[ID["TL"]] += py_itf8(int(c["tag_list"][r]))
        b0 = int(s["base_off"][r])
        rb, rq = s["bases"][b0:b0 + L].tobytes(), s["quals"][b0:b0 + L].tobytes()
        if not int(c["bam_flags"][r]) & 4:
            f0, f1 = int(f["off"][r]), int(f["off"][r + 1])
            out[ID["FN"]] += py_itf8(f1 - f0)
            pp = 0
            for k in range(f0, f1):
                o, p, a, b = int(f["op"][k]), int(f["pos"][k]), int(f["a"][k]), int(f["b"][k])
                out[ID["FC"]].append(o)
                out[ID["FP"]] += py_itf8(p - pp)
                pp = p
                ch = chr(o)
                if ch == "B":
                    out[ID["BA"]].append(a & 255)
                    out[ID["QS"]].append(b & 255)
                elif ch == "X":
                    out[ID["BS"]].append((a >> 16) & 255)
                elif ch in "IS":
                    out[ID["IN" if ch == "I" else "SC"]] += rb[a:a + b] + b"\t"
                elif ch in "HPDN":
                    out[ID[dict(H="HC", P="PD", D="DL", N="RS")[ch]]] += py_itf8(a)
                elif ch == "i":
                    out[ID["BA"]].append(a & 255)
                elif ch == "Q":
                    out[ID["QS"]].append(a & 255)
            out[ID["MQ"]] += py_itf8(int(s["mapq"][r]))
            if cf & 1:
                out[ID["QS"]] += rq
        else:
            if not cf & 8:
                out[ID["BA"]] += rb
            if cf & 1:
                out[ID["QS"]] += rq
    return [bytes(x) for x in out]
