"""Random batches of records for the container-assembly / BAM tests (tests/test_oracle_bam.py, tests/test_bam_gpu.py)
and a plain-Python statement of one BAM alignment record as the SAM specification §4.2 lays it out."""
import struct

import numpy as np

from tests import series_cases as sc

NYBBLE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
NYBBLE.update({c.lower(): i for c, i in list(NYBBLE.items()) if c.isalpha()})
NYBBLE["."] = 15


def reg2bin(beg, end):
    """SAM specification §5.3 (0-based, half-open)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def py_bam_record(ref_id, pos1, mapq, bin_, flag, name, cigar_words, seq, qual, mate_ref, mate_pos1, tlen, tags=b""):
    """block_size + the record of SAM specification §4.2: every field little-endian, seq 4 bits per base (high nybble
    first), qual raw bytes (0xFF x l_seq when absent)."""
    L = len(seq)
    packed = bytearray((L + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= NYBBLE[c] << (0 if i & 1 else 4)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos1 - 1, len(name) + 1, mapq, bin_, len(cigar_words), flag, L, mate_ref,
                       mate_pos1 - 1, tlen)
    body += name.encode() + b"\0" + b"".join(struct.pack("<I", w) for w in cigar_words) + bytes(packed)
    body += bytes(qual) if qual is not None else b"\xff" * L
    body += tags
    return struct.pack("<i", len(body)) + body


def random_batch(rng, sizes, max_len=60):
    """Containers of random records (series_cases.random_slice) with next-fragment distances that stay inside their
    container (a few records point at the same mate, a few point outside), lengths that fit 32-bit positions, a sequence
    id per container and read names / tags for every record."""
    parts = [sc.random_slice(rng, n, max_len=max_len) for n in sizes]
    n = sum(sizes)
    cat = lambda f: np.concatenate([f(p) for p in parts])
    cols = {k: cat(lambda p: p["cols"][k]).astype(np.int32) for k in sc_cols()}
    read_len = cat(lambda p: p["read_len"])
    base_off = np.concatenate([[0], np.cumsum(read_len)]).astype(np.int64)
    bases = np.concatenate([p["bases"][:int(p["base_off"][-1])] for p in parts] + [np.zeros(16, np.uint8)])
    quals = np.concatenate([p["quals"][:int(p["base_off"][-1])] for p in parts] + [np.zeros(16, np.uint8)])
    f_off = np.concatenate([[0], np.cumsum(cat(lambda p: np.diff(p["feats"]["off"])))]).astype(np.int64)
    feats = dict(off=f_off, **{k: cat(lambda p: p["feats"][k]) for k in ("op", "pos", "a", "b")})
    big = np.isin(feats["op"], np.frombuffer(b"HPDN", np.uint8))
    feats["a"] = np.where(big, feats["a"] % (1 << 18), feats["a"]).astype(np.int32)
    cont_first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cont = np.repeat(np.arange(len(sizes)), sizes)
    # distances: inside the container for most records that claim a mate downstream
    for r in range(n):
        last = int(cont_first[cont[r] + 1]) - 1
        room = last - r - 1  # largest valid distance
        roll = rng.random()
        if room < 0 or roll < 0.03:
            cols["next_frag"][r] = int(rng.integers(room + 1, room + 50))  # outside: the reference throws
        elif roll < 0.06:
            cols["next_frag"][r] = room  # many records point at the last one: shared mates
        else:
            cols["next_frag"][r] = int(rng.integers(0, min(room, 6) + 1))
    cols["bam_flags"] = (cols["bam_flags"] | np.where(rng.random(n) < 0.7, 1, 0)).astype(np.int32)  # mostly paired
    multiref = (rng.random(len(sizes)) < 0.4).astype(np.uint8)
    cont_ref_id = rng.integers(-1, 25, len(sizes)).astype(np.int32)
    ref_id = np.where(multiref[cont] != 0, cols["ref_id"], cont_ref_id[cont]).astype(np.int32)
    names = [("q%d_%d" % (r, int(rng.integers(0, 10 ** int(rng.integers(1, 9)))))).encode() for r in range(n)]
    tags = [bytes(rng.integers(0, 256, int(rng.integers(0, 40))).astype(np.uint8)) for r in range(n)]
    name_off = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
    tag_off = np.concatenate([[0], np.cumsum([len(x) for x in tags])]).astype(np.int64)
    return dict(n=n, sizes=list(sizes), cont_first=cont_first, cont=cont, cols=cols, read_len=read_len.astype(np.int32),
                aln_start=cat(lambda p: p["aln_start"]).astype(np.int32), mapq=cat(lambda p: p["mapq"]).astype(np.uint8),
                base_off=base_off, bases=bases, quals=quals, feats=feats, multiref=multiref, cont_ref_id=cont_ref_id, ref_id=ref_id,
                names=np.frombuffer(b"".join(names) + b"\0", np.uint8), name_off=name_off,
                tags=np.frombuffer(b"".join(tags) + b"\0", np.uint8), tag_off=tag_off)


def sc_cols():
    return ("bam_flags", "comp_flags", "ref_id", "read_group", "mate_flags", "mate_ref", "mate_start", "tlen", "next_frag",
            "tag_list")


def insert_size(bf, ref, start, end, a, z):
    """CramNormalizer.java:455-470 in plain Python (records a, z)."""
    if (bf[a] & 4) or (bf[z] & 4) or ref[a] != ref[z]:
        return 0
    p1 = end[a] if bf[a] & 16 else start[a]
    p2 = end[z] if bf[z] & 16 else start[z]
    return p2 - p1 + (1 if p2 >= p1 else -1)


def mate_batch(rng, sizes):
    """Containers of records as Sam2CramRecordFactory leaves them before setMateInfo (CramSerilization.java:274-373):
    templates of one record (paired or not), proper pairs (every field consistent), pairs with one field off, templates of
    three and four records, secondary alignments that share the name of a primary pair; the records of a container in
    random order; the same names in every container; names of different lengths that share prefixes."""
    cols = {k: [] for k in sc_cols()}
    names, start, end, ref = [], [], [], []
    want_kind = []
    for ci, n in enumerate(sizes):
        recs = []  # (template id, kind)
        t = 0
        while len(recs) < n:
            room = n - len(recs)
            k = rng.choice(["single", "lone", "pair", "pair", "pair", "bad", "three", "four", "sec"])
            need = dict(single=1, lone=1, pair=2, bad=2, three=3, four=4, sec=4)[k]
            if need > room:
                k, need = "lone", 1
            recs += [(t, k, j) for j in range(need)]
            t += 1
        order = rng.permutation(len(recs))
        recs = [recs[i] for i in order]
        base = len(names)
        for (tid, k, j) in recs:
            names.append(("t%d" % tid).encode() * (1 + tid % 3) + (b"/x" if tid % 5 == 0 else b""))
            paired = k != "single"
            bf = (1 if paired else 0) | (16 if rng.random() < 0.5 else 0) | (4 if rng.random() < 0.05 else 0)
            if k == "sec" and j >= 2:
                bf |= 0x100
            cols["bam_flags"].append(bf)
            cols["comp_flags"].append(int(rng.integers(0, 16)))
            cols["next_frag"].append(int(rng.integers(-1, 5)))
            cols["ref_id"].append(int(rng.integers(0, 3)) if rng.random() < 0.2 else 1)
            s = int(rng.integers(1, 5000))
            start.append(s)
            end.append(s + int(rng.integers(0, 200)))
            ref.append(cols["ref_id"][-1])
            for c in ("mate_flags", "mate_ref", "mate_start", "tlen"):
                cols[c].append(int(rng.integers(0, 3)))
            cols["read_group"].append(0)
            cols["tag_list"].append(0)
        # make the two-record templates consistent in container order, then spoil the "bad" ones
        groups = {}
        for i, (tid, k, j) in enumerate(recs):
            groups.setdefault((tid, bool(cols["bam_flags"][base + i] & 0x100)), []).append(base + i)
        for (tid, sec), idx in groups.items():
            kind = recs[idx[0] - base][1]
            if len(idx) != 2 or kind == "single":
                continue
            a, z = idx
            fl = (0x40, 0x80) if rng.random() < 0.5 else (0x80, 0x40)
            cols["bam_flags"][a] |= fl[0]
            cols["bam_flags"][z] |= fl[1]
            bfv = cols["bam_flags"]
            tl = insert_size(bfv, ref, start, end, a, z)
            cols["tlen"][a], cols["tlen"][z] = tl, -tl
            cols["mate_ref"][a], cols["mate_ref"][z] = ref[z], ref[a]
            cols["mate_start"][a], cols["mate_start"][z] = start[z], start[a]
            cols["mate_flags"][a] = (2 if rng.random() < 0.3 else 0) | (1 if bfv[z] & 16 else 0)
            cols["mate_flags"][z] = (2 if rng.random() < 0.3 else 0) | (1 if bfv[a] & 16 else 0)
            if kind == "bad":
                w = int(rng.integers(0, 7))
                e = a if rng.random() < 0.5 else z
                if w == 0:
                    cols["tlen"][e] += 1
                elif w == 1:
                    cols["mate_start"][e] += 1
                elif w == 2:
                    cols["mate_ref"][e] += 1
                elif w == 3:
                    cols["mate_flags"][e] ^= 1
                elif w == 4:
                    cols["bam_flags"][a] &= ~0xC0  # neither first nor last
                elif w == 5:
                    cols["bam_flags"][z] = (cols["bam_flags"][z] & ~0xC0) | (cols["bam_flags"][a] & 0xC0)  # both the same end
                else:
                    cols["tlen"][a], cols["tlen"][z] = cols["tlen"][z], cols["tlen"][a]
                    if cols["tlen"][a] == 0:
                        cols["tlen"][a] = 5
        want_kind += [r[1] for r in recs]
    cols = {k: np.asarray(v, np.int32) for k, v in cols.items()}
    name_off = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
    return dict(n=len(names), cont_first=np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), cols=cols,
                aln_start=np.asarray(start, np.int32), aln_end=np.asarray(end, np.int32),
                names=np.frombuffer(b"".join(names) + b"\0", np.uint8), name_off=name_off, kind=want_kind)


# ---- the reference's own BAM-record test inputs ----
# src/test/java/net/sf/cram/TestBAMRecordView.java:47-66 builds ONE record through BAMRecordView: refID 0, start 77,
# mapping score 44, index bin 99, flags 555, mate 0 / 78, insert size 133, "name1", 10M, ten 'A', ten 'B' (raw bytes: the
# test passes "BBBBBBBBBB".getBytes()), tag AM:A:Q.  The test prints the bytes and compares nothing; the bytes below are
# worked from src/main/java/htsjdk/samtools/cram/encoding/reader/BAMRecordView.java (the layout py_bam_record states on its
# own).  The bin 99 is a value the test sets by hand; an encoder computes reg2bin of the alignment (BAMRecordCodec#encode).
REF_VIEW_RECORD = dict(ref_id=0, start=77, end=86, mapq=44, bin_set_by_the_test=99, flags=555, mate_ref=0, mate_start=78, tlen=133,
                       name="name1", cigar=[(10 << 4) | 0], seq="A" * 10, qual=b"B" * 10, tags=b"AMAQ")
# src/test/java/htsjdk/samtools/cram/encoding/reader/BAMRecordViewTest.java:78-124 writes an unmapped record twice
# ("readName", then "2") through its translate() (:35-49, which never sets the index bin: it stays 0) and reads both back
# with BAMRecordCodec: flags 4, no reference, start 0, mapping quality 0, empty CIGAR, no mate, base "A", score '!' = 0.
REF_UNMAPPED_PAIR = [dict(name="readName"), dict(name="2")]


def ref_view_record_bytes(bin_):
    r = REF_VIEW_RECORD
    return py_bam_record(r["ref_id"], r["start"], r["mapq"], bin_, r["flags"], r["name"], r["cigar"], r["seq"], r["qual"],
                         r["mate_ref"], r["mate_start"], r["tlen"], r["tags"])


def ref_unmapped_pair_bytes():
    return b"".join(py_bam_record(-1, 0, 0, 0, 4, r["name"], [], "A", b"\0", -1, 0, 0) for r in REF_UNMAPPED_PAIR)
