"""Host-side mirror of the reference's record-level interfaces over flat containers, backed by libcramgpu (HIP only).

Reference interface (SURVEY.md §8(b) level 2)                          -> here
  htsjdk!build/Sam2CramRecordFactory#setRefBases / #createCramRecord     -> Sam2CramRecordFactory.set_ref_bases / create_variations
  htsjdk!structure/SubstitutionMatrix#<init>([[J), #code                 -> substitution_matrix / assign_codes
  CramNormalizer.restoreReadBases / restoreQualityScores (CramNormalizer.java:335-441, :287-333)
                                                                         -> CramNormalizer.restore_read_bases / restore_quality_scores
  QualityScorePreservation(spec).addQualityScores (QualityScorePreservation.java:139-301)
                                                                         -> QualityScorePreservation(spec).add_quality_scores
The object graph of the reference (SAMRecord, CramCompressionRecord, ReadFeature lists) is replaced by flat arrays
(`ReadBatch`, `FeatureBatch`) because that is what crosses the JNI boundary; semantics and error behaviour follow the
reference method by method.  Everything runs on the GPU; there is no CPU implementation in this package.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CramGpuError, Features, Policy, Reads, RefWin

RF_UNMAPPED, RF_NULL_QUALS, RF_NO_BASES = _lib.RF_UNMAPPED, _lib.RF_NULL_QUALS, _lib.RF_NO_BASES
RF_UNKNOWN_BASES, RF_FORCE_QS, RF_SKIP = _lib.RF_UNKNOWN_BASES, _lib.RF_FORCE_QS, _lib.RF_SKIP
DEFAULT_QUALITY_SCORE = 30  # '?' - '!' (CramNormalizer.java:141,204)

_CIGAR_OPS = "MIDNSHP=X"


def pack_cigar(text):
    out, num = [], 0
    for ch in text:
        if ch.isdigit():
            num = num * 10 + int(ch)
        else:
            out.append((num << 4) | _CIGAR_OPS.index(ch))
            num = 0
    return out


def _torch():
    import torch
    return torch


def _dev(x, dtype, device):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(device)


def _stream():
    return _torch().cuda.current_stream().cuda_stream


class ReadBatch:
    """Reads of one or more containers as device arrays (the cramgpu_reads struct of include/cramgpu.h)."""

    def __init__(self, device, bases, base_off, read_len, aln_start, cigar, cigar_off, n_cigar, quals=None, mapq=None,
                 rflags=None, read_cont=None, windows=((0, 0, 0),)):
        torch = _torch()
        self.device = device
        n = len(read_len)
        self.n = n
        self.bases = _dev(bases, torch.uint8, device)
        self.quals = _dev(quals, torch.uint8, device) if quals is not None else torch.zeros_like(self.bases)
        self.base_off = _dev(base_off, torch.int64, device)
        self.read_len = _dev(read_len, torch.int32, device)
        self.aln_start = _dev(aln_start, torch.int32, device)
        self.cigar = _dev(cigar, torch.int32, device)  # bit pattern of uint32
        self.cigar_off = _dev(cigar_off, torch.int64, device)
        self.n_cigar = _dev(n_cigar, torch.int32, device)
        self.mapq = _dev(mapq if mapq is not None else np.full(n, 60, np.uint8), torch.uint8, device)
        self.rflags = _dev(rflags if rflags is not None else np.zeros(n, np.uint8), torch.uint8, device)
        self.read_cont = _dev(read_cont, torch.int32, device) if read_cont is not None else None
        w = np.asarray(windows, np.int64).reshape(-1, 3)  # (ref_off, ref_len, ref_start0) per container
        self.n_cont = w.shape[0]
        self.win = _dev(w.reshape(-1), torch.int64, device)
        self.max_read_len = int(self.read_len.max().item()) if n else 0
        self._keep = None

    @staticmethod
    def from_lists(device, starts, cigars, bases, quals=None, mapq=None, rflags=None, ref_len=0):
        """Small batches from Python lists (tests): bases/quals are lists of bytes, cigars strings or packed lists."""
        lens = [len(b) for b in bases]
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if lens else np.zeros(0, np.int64)
        packed = [pack_cigar(c) if isinstance(c, str) else list(c) for c in cigars]
        nc = [len(c) for c in packed]
        coff = np.concatenate([[0], np.cumsum(nc)[:-1]]).astype(np.int64) if nc else np.zeros(0, np.int64)
        flat = np.asarray([x for c in packed for x in c] + [0], np.uint32).view(np.int32)
        b = np.frombuffer(b"".join(bytes(x) for x in bases) + b"\0" * 16, np.uint8)
        q = np.frombuffer(b"".join(bytes(x) for x in quals) + b"\0" * 16, np.uint8) if quals is not None else None
        return ReadBatch(device, b, off, lens, starts, flat, coff, nc, quals=q, mapq=mapq, rflags=rflags,
                         windows=((0, ref_len, 0),))

    def struct(self):
        r = Reads()
        r.n_reads, r.n_cont = self.n, self.n_cont
        r.n_bases, r.max_read_len = int(self.bases.numel()), self.max_read_len
        r.bases, r.quals = self.bases.data_ptr(), self.quals.data_ptr()
        r.base_off, r.read_len = self.base_off.data_ptr(), self.read_len.data_ptr()
        r.aln_start, r.rflags, r.mapq = self.aln_start.data_ptr(), self.rflags.data_ptr(), self.mapq.data_ptr()
        r.cigar, r.cigar_off, r.n_cigar = self.cigar.data_ptr(), self.cigar_off.data_ptr(), self.n_cigar.data_ptr()
        r.read_cont = self.read_cont.data_ptr() if self.read_cont is not None else None
        r.win = self.win.data_ptr()
        return r


class FeatureBatch:
    """Flat read features on the device (cramgpu_features)."""

    def __init__(self, device, n_reads, cap):
        torch = _torch()
        cap = max(int(cap), 1)
        self.cap = cap
        self.n_reads = n_reads
        self.op = torch.zeros(cap, dtype=torch.uint8, device=device)
        self.pos = torch.zeros(cap, dtype=torch.int32, device=device)
        self.a = torch.zeros(cap, dtype=torch.int32, device=device)
        self.b = torch.zeros(cap, dtype=torch.int32, device=device)
        self.off = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
        self.count = 0

    @staticmethod
    def from_host(device, op, pos, a, b, off):
        f = FeatureBatch(device, len(off) - 1, len(op))
        torch = _torch()
        if len(op):
            f.op[:len(op)] = torch.as_tensor(np.ascontiguousarray(op, np.uint8))
            f.pos[:len(op)] = torch.as_tensor(np.ascontiguousarray(pos, np.int32))
            f.a[:len(op)] = torch.as_tensor(np.ascontiguousarray(a, np.int32))
            f.b[:len(op)] = torch.as_tensor(np.ascontiguousarray(b, np.int32))
        f.off.copy_(torch.as_tensor(np.ascontiguousarray(off, np.int64)))
        f.count = len(op)
        return f

    def struct(self):
        s = Features()
        s.op, s.pos, s.a, s.b, s.off = (self.op.data_ptr(), self.pos.data_ptr(), self.a.data_ptr(), self.b.data_ptr(),
                                        self.off.data_ptr())
        s.cap = self.cap
        return s

    def to_host(self):
        n = self.count
        return dict(op=self.op[:n].cpu().numpy(), pos=self.pos[:n].cpu().numpy(), a=self.a[:n].cpu().numpy(),
                    b=self.b[:n].cpu().numpy(), off=self.off.cpu().numpy())


RECORD_COLS = ("bam_flags", "comp_flags", "ref_id", "read_group", "mate_flags", "mate_ref", "mate_start", "tlen", "next_frag",
               "tag_list")


class RecordColumns:
    """The CramCompressionRecord fields that reads and features do not hold, as device int32 arrays
    (cramgpu_record_cols): what htsjdk!encoding/writer/Writer#write reads besides bases, scores and features."""

    def __init__(self, device, n_reads, multiref=None, ap_base=None, ap_delta=True, alias_qs=False, names=None, name_off=None,
                 name_len=None, capture_names=False, **cols):
        """names / name_off / name_len: device tensors (uint8 bytes the names lie in, int64 [n], int32 [n]) — the writer
        then fills RN (every record's name with capture_names, the detached records' otherwise)."""
        torch = _torch()
        self.device, self.n = device, n_reads
        self.names, self.name_off, self.name_len, self.capture_names = names, name_off, name_len, bool(capture_names)
        for k in RECORD_COLS:
            v = cols.get(k)
            setattr(self, k, _dev(v, torch.int32, device) if v is not None else torch.zeros(n_reads, dtype=torch.int32, device=device))
        self.multiref = _dev(multiref, torch.uint8, device) if multiref is not None else None
        self.ap_base = _dev(ap_base, torch.int32, device) if ap_base is not None else None
        self.ap_delta = bool(ap_delta)
        self.alias_qs = bool(alias_qs)

    def struct(self):
        c = _lib.RecordCols()
        for k in RECORD_COLS:
            setattr(c, k, getattr(self, k).data_ptr())
        c.cont_multiref = self.multiref.data_ptr() if self.multiref is not None else None
        c.cont_ap_base = self.ap_base.data_ptr() if self.ap_base is not None else None
        c.ap_delta = 1 if self.ap_delta else 0
        c.flags = _lib.WS_ALIAS_QS if getattr(self, "alias_qs", False) else 0
        if getattr(self, "names", None) is not None:
            c.names, c.name_off, c.name_len = self.names.data_ptr(), self.name_off.data_ptr(), self.name_len.data_ptr()
        c.capture_names = 1 if getattr(self, "capture_names", False) else 0
        return c


class Writer:
    """htsjdk!encoding/writer/Writer (L139-261) over whole containers: every record's values appended to its slice's
    data series, one byte stream per content id (cramgpu_write_series_dev).  The streams of the batch lie in `arena`
    (device), 16-byte aligned, at off[c, id] with length len[c, id]: the input arena of the rANS encoder as it is."""

    def __init__(self, engine, device, n_cont, arena_bytes):
        torch = _torch()
        self.engine, self.device, self.n_cont = engine, device, n_cont
        self.arena = torch.zeros(int(arena_bytes) + 16, dtype=torch.uint8, device=device)
        self.off = torch.zeros(n_cont * _lib.N_SERIES, dtype=torch.int64, device=device)
        self.len = torch.zeros(n_cont * _lib.N_SERIES, dtype=torch.int32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)

    def write(self, reads, cols, feats):
        """Stream-ordered; lengths() synchronises."""
        rs, cs, fs = reads.struct(), cols.struct(), feats.struct()
        _lib.check(_lib.lib().cramgpu_write_series_dev(self.engine.handle, C.byref(rs), C.byref(cs), C.byref(fs),
                                                       self.arena.data_ptr(), int(self.arena.numel()), self.off.data_ptr(),
                                                       self.len.data_ptr(), self.status.data_ptr(), _stream()))

    def lengths(self):
        """(off[n_cont, 30], len[n_cont, 30]) on the host; raises what the device reported."""
        st = int(self.status.item())
        if st != 0:
            raise _lib.CramGpuError(st, "data-series writer")
        return (self.off.cpu().numpy().reshape(self.n_cont, _lib.N_SERIES), self.len.cpu().numpy().reshape(self.n_cont, _lib.N_SERIES))

    def series(self, c, name_or_id):
        """bytes of one stream (tests)."""
        k = _lib.SERIES_ID[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        off, ln = self.lengths()
        return self.arena[int(off[c, k]):int(off[c, k]) + int(ln[c, k])].cpu().numpy().tobytes()


class CramRecordReader:
    """htsjdk!encoding/reader/CramRecordReader#read (L59-211) over whole containers (cramgpu_read_series_dev): the
    streams of every series back into record columns, features, the bytes of I / S features, the bases of unmapped
    records and the quality arrays (0xFF where a record has none: what CramNormalizer.restore_quality_scores fills)."""

    def __init__(self, engine, device, n_reads, n_cont, bases_cap, feature_cap, seq_cap):
        torch = _torch()
        self.engine, self.device, self.n, self.n_cont = engine, device, n_reads, n_cont
        self.cols = RecordColumns(device, n_reads)
        self.read_len = torch.zeros(n_reads, dtype=torch.int32, device=device)
        self.aln_start = torch.zeros(n_reads, dtype=torch.int32, device=device)
        self.mapq = torch.zeros(n_reads, dtype=torch.uint8, device=device)
        self.rflags = torch.zeros(n_reads, dtype=torch.uint8, device=device)
        self.base_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
        self.bases = torch.zeros(int(bases_cap) + 16, dtype=torch.uint8, device=device)
        self.quals = torch.zeros(int(bases_cap) + 16, dtype=torch.uint8, device=device)
        self.seq = torch.zeros(max(int(seq_cap), 1) + 16, dtype=torch.uint8, device=device)
        self.seq_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
        self.feats = FeatureBatch(device, n_reads, feature_cap)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.bases_cap, self.seq_cap = int(bases_cap), max(int(seq_cap), 1)
        # read names stay in the RN stream: offset (relative to the arena) and length per record, -1 = the record has none
        self.name_off = torch.zeros(n_reads, dtype=torch.int64, device=device)
        self.name_len = torch.full((n_reads,), -1, dtype=torch.int32, device=device)

    def read(self, arena, series_off, series_len, cont_first, series_bytes, multiref=None, ap_base=None, ap_delta=True,
             capture_names=False):
        """arena / series_off / series_len / cont_first: device tensors (cont_first int32 [n_cont + 1]).  Stream-ordered."""
        torch = _torch()
        self.cols.capture_names = bool(capture_names)
        self.cols.multiref = _dev(multiref, torch.uint8, self.device) if multiref is not None else None
        self.cols.ap_base = _dev(ap_base, torch.int32, self.device) if ap_base is not None else None
        self.cols.ap_delta = bool(ap_delta)
        cs, fs, o = self.cols.struct(), self.feats.struct(), _lib.ReadOut()
        o.read_len, o.aln_start, o.mapq, o.base_off = (self.read_len.data_ptr(), self.aln_start.data_ptr(), self.mapq.data_ptr(),
                                                       self.base_off.data_ptr())
        o.bases, o.quals, o.bases_cap = self.bases.data_ptr(), self.quals.data_ptr(), self.bases_cap
        o.seq, o.seq_off, o.seq_cap = self.seq.data_ptr(), self.seq_off.data_ptr(), self.seq_cap
        o.rflags = self.rflags.data_ptr()
        o.name_off, o.name_len = self.name_off.data_ptr(), self.name_len.data_ptr()
        self._keep = (cont_first, series_off, series_len)
        _lib.check(_lib.lib().cramgpu_read_series_dev(self.engine.handle, self.n, self.n_cont, cont_first.data_ptr(),
                                                      arena if isinstance(arena, int) else arena.data_ptr(),
                                                      series_off.data_ptr(), series_len.data_ptr(), int(series_bytes), C.byref(cs),
                                                      C.byref(o), C.byref(fs), self.status.data_ptr(), _stream()))

    def check(self):
        """Synchronises; raises what the device reported; sets feats.count."""
        st = int(self.status.item())
        if st != 0:
            raise _lib.CramGpuError(st, "data-series reader")
        self.feats.count = int(self.feats.off[self.n].item())
        return self


def substitution_matrix(hist25):
    """SubstitutionMatrix#<init>([[J): 25 counts ([ref][base] over ACGTN) -> 5 header bytes."""
    h = np.ascontiguousarray(hist25, np.int64).reshape(-1)
    assert h.size == 25
    m = np.zeros(5, np.uint8)
    _lib.lib().cramgpu_substitution_matrix(h.ctypes.data, m.ctypes.data)
    return m


class Sam2CramRecordFactory:
    """htsjdk!build/Sam2CramRecordFactory: reference-diff read features for a batch of reads."""

    def __init__(self, engine):
        self.engine = engine
        self.ref = None

    def set_ref_bases(self, d_ref):
        """setRefBases(byte[]): the reference arena the batch's windows point into (device uint8 tensor)."""
        self.ref = d_ref

    def create_variations(self, reads, feature_cap=None, with_hist=True):
        """createVariations for every read.  Returns (FeatureBatch, sub_hist[n_cont][25] on the host or None)."""
        torch = _torch()
        cap = feature_cap if feature_cap is not None else reads.n * 8 + 1024
        while True:
            feats = FeatureBatch(reads.device, reads.n, cap)
            hist = torch.zeros(reads.n_cont * 25, dtype=torch.int64, device=reads.device) if with_hist else None
            rs, fs = reads.struct(), feats.struct()
            n = _lib.lib().cramgpu_extract_features_dev(self.engine.handle, C.byref(rs), self.ref.data_ptr(), C.byref(fs),
                                                        hist.data_ptr() if with_hist else None, _stream())
            if n == _lib.E_INSUFFICIENT:
                cap = int(feats.off[-1].item()) + 16  # the scan already knows the exact need
                continue
            _lib.check(n)
            feats.count = int(n)
            torch.cuda.synchronize()
            return feats, (hist.cpu().numpy().reshape(reads.n_cont, 25) if with_hist else None)

    def assign_codes(self, reads, feats, matrices):
        """Writer#write: code = matrix.code(refBase, base) for every Substitution feature."""
        m = np.ascontiguousarray(matrices, np.uint8).reshape(-1)
        fs = feats.struct()
        _lib.check(_lib.lib().cramgpu_assign_codes_dev(
            self.engine.handle, reads.n, reads.read_cont.data_ptr() if reads.read_cont is not None else None,
            reads.n_cont, C.byref(fs), m.ctypes.data, _stream()))


    def matrices_and_codes(self, reads, feats, d_hist, d_matrices):
        """F3 + code assignment on the device (no host round trip): d_hist int64[n_cont*25] from create_variations,
        d_matrices uint8[n_cont*5] receives the SubstitutionMatrix header bytes of every container."""
        fs = feats.struct()
        _lib.check(_lib.lib().cramgpu_matrices_assign_codes_dev(
            self.engine.handle, reads.n, reads.read_cont.data_ptr() if reads.read_cont is not None else None,
            reads.n_cont, C.byref(fs), d_hist.data_ptr(), d_matrices.data_ptr(), _stream()))


class CramNormalizer:
    """src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java: base and quality restoration (decode side)."""

    def __init__(self, engine):
        self.engine = engine

    def restore_read_bases(self, device, read_len, aln_start, rflags, feats, d_ref, windows, matrices, seq=None,
                           seq_off=None, read_cont=None):
        torch = _torch()
        n = len(read_len)
        rl = _dev(read_len, torch.int32, device)
        st = _dev(aln_start, torch.int32, device)
        fl = _dev(rflags, torch.uint8, device)
        w = np.asarray(windows, np.int64).reshape(-1, 3)
        d_win = _dev(w.reshape(-1), torch.int64, device)
        out_off = torch.cumsum(rl.to(torch.int64), 0) - rl.to(torch.int64)
        out = torch.zeros(int(rl.sum().item()) + 16, dtype=torch.uint8, device=device)
        status = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        d_seq = _dev(seq, torch.uint8, device) if seq is not None else torch.zeros(16, dtype=torch.uint8, device=device)
        d_seq_off = _dev(seq_off, torch.int64, device) if seq_off is not None else torch.zeros(max(n, 1), dtype=torch.int64,
                                                                                              device=device)
        rc = _dev(read_cont, torch.int32, device) if read_cont is not None else None
        m = np.ascontiguousarray(matrices, np.uint8).reshape(-1)
        fs = feats.struct()
        _lib.check(_lib.lib().cramgpu_restore_bases_dev(
            self.engine.handle, n, rl.data_ptr(), st.data_ptr(), fl.data_ptr(), rc.data_ptr() if rc is not None else None,
            w.shape[0], d_win.data_ptr(), C.byref(fs), d_seq.data_ptr(), d_seq_off.data_ptr(), int(d_seq.numel()),
            d_ref.data_ptr(),
            m.ctypes.data, out.data_ptr(), out_off.data_ptr(), status.data_ptr(), _stream()))
        torch.cuda.synchronize()
        return out, out_off, status[:n]

    def restore_quality_scores(self, device, read_len, rflags, feats, scores, default_qs=DEFAULT_QUALITY_SCORE):
        torch = _torch()
        n = len(read_len)
        rl = _dev(read_len, torch.int32, device)
        fl = _dev(rflags, torch.uint8, device)
        out_off = torch.cumsum(rl.to(torch.int64), 0) - rl.to(torch.int64)
        sc = _dev(scores, torch.uint8, device).clone()
        status = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        fs = feats.struct()
        _lib.check(_lib.lib().cramgpu_restore_quals_dev(self.engine.handle, n, rl.data_ptr(), fl.data_ptr(), C.byref(fs),
                                                        int(default_qs), sc.data_ptr(), out_off.data_ptr(),
                                                        status.data_ptr(), _stream()))
        torch.cuda.synchronize()
        return sc, status[:n]


def parse_policies(spec):
    """QualityScorePreservation.parsePolicies: spec -> ctypes Policy array sorted like the reference (host only)."""
    arr = (Policy * 64)()
    n = _lib.check(_lib.lib().cramgpu_parse_policies(spec.encode(), arr, 64))
    return arr, n


class QualityScorePreservation:
    """src/main/java/htsjdk/samtools/cram/lossy/QualityScorePreservation.java (lossy quality model)."""

    WINDOW = 1000000  # ReferenceTracks.java:23

    def __init__(self, engine, spec):
        self.engine = engine
        self.spec = spec
        self.policies, self.n_pol = parse_policies(spec)

    def tracks_windows(self, aln_start, aln_span, rflags, mapq, ref_len, window=None):
        """ReferenceTracks.ensureRange replayed in read order (host): window position per read."""
        n = len(aln_start)
        a = np.ascontiguousarray(aln_start, np.int32)
        sp = np.ascontiguousarray(aln_span, np.int32)
        fl = np.ascontiguousarray(rflags, np.uint8)
        mq = np.ascontiguousarray(mapq, np.uint8)
        wpos = np.zeros(max(n, 1), np.int32)
        at = C.c_int32(-1)
        r = _lib.lib().cramgpu_tracks_windows(n, a.ctypes.data, sp.ctypes.data, fl.ctypes.data, mq.ctypes.data,
                                              self.policies, self.n_pol, int(ref_len),
                                              int(window if window is not None else self.WINDOW), wpos.ctypes.data,
                                              C.byref(at))
        if r == _lib.E_JAVA_THROWS:
            raise CramGpuError(r, "ReferenceTracks would throw at read %d" % at.value)
        _lib.check(r)
        return wpos[:n]

    def add_quality_scores(self, reads, d_ref, wpos, window=None):
        """addQualityScores for every read: (scores uint8 with 0xFF holes, force flags, status)."""
        torch = _torch()
        scores = torch.zeros(reads.bases.numel(), dtype=torch.uint8, device=reads.device)
        force = torch.zeros(max(reads.n, 1), dtype=torch.uint8, device=reads.device)
        status = torch.zeros(max(reads.n, 1), dtype=torch.int32, device=reads.device)
        d_wpos = _dev(wpos, torch.int32, reads.device)
        rs = reads.struct()
        _lib.check(_lib.lib().cramgpu_lossy_quals_dev(self.engine.handle, C.byref(rs), d_ref.data_ptr(), d_wpos.data_ptr(),
                                                      int(window if window is not None else self.WINDOW), self.policies,
                                                      self.n_pol, scores.data_ptr(), force.data_ptr(), status.data_ptr(),
                                                      _stream()))
        torch.cuda.synchronize()
        return scores, force[:reads.n], status[:reads.n]

    def merge_quality_features(self, reads, scores, force, status, feats):
        """Tail of addQualityScores: BaseQualityScore features for non-forced reads, stable-sorted by position."""
        torch = _torch()
        cap = feats.count + int(reads.bases.numel()) + 16
        out = FeatureBatch(reads.device, reads.n, cap)
        fi, fo = feats.struct(), out.struct()
        n = _lib.check(_lib.lib().cramgpu_merge_quality_features_dev(
            self.engine.handle, reads.n, reads.read_len.data_ptr(), reads.base_off.data_ptr(), scores.data_ptr(),
            force.data_ptr(), status.data_ptr(), C.byref(fi), C.byref(fo), _stream()))
        out.count = int(n)
        torch.cuda.synchronize()
        return out


# ======================================================================================================
# Host form of level 2 (cramgpu_submit_reads / cramgpu_submit_restore): numpy arrays in, numpy arrays out, no torch.
# This is the call sequence of the JNI shim of INTEGRATION.md, written in Python because the image has no JVM.
# ======================================================================================================
def _np(x, dtype):
    return np.ascontiguousarray(x, dtype)


def ref_upload(engine, contig, bases):
    """ReferenceSource.getReferenceBases -> Sam2CramRecordFactory#setRefBases: one contig's bases to the GPU, once."""
    b = _np(np.frombuffer(bytes(bases), np.uint8) if isinstance(bases, (bytes, bytearray)) else bases, np.uint8)
    _lib.check(_lib.lib().cramgpu_ref_upload(engine.handle, int(contig), b.ctypes.data, b.size))


class HostReadJob:
    """One outstanding cramgpu_submit_reads: keeps the host arrays alive until wait()."""

    def __init__(self, engine, starts, cigars, bases, cont_reads, cont_contig, quals=None, mapq=None, rflags=None,
                 lossy_spec=None, window=0, feat_cap=None):
        n = len(starts)
        packed = [pack_cigar(c) if isinstance(c, str) else list(c) for c in cigars]
        self.read_len = _np([len(b) for b in bases], np.int32)
        self.bases = _np(np.frombuffer(b"".join(bytes(b) for b in bases) + b"\0" * 16, np.uint8), np.uint8)
        self.quals = (_np(np.frombuffer(b"".join(bytes(q) for q in quals) + b"\0" * 16, np.uint8), np.uint8)
                      if quals is not None else None)
        self.aln_start = _np(starts, np.int32)
        self.rflags = _np(rflags if rflags is not None else np.zeros(n), np.uint8)
        self.mapq = _np(mapq if mapq is not None else np.full(n, 60), np.uint8)
        self.n_cigar = _np([len(c) for c in packed], np.int32)
        self.cigar = _np([x for c in packed for x in c] + [0], np.uint32)
        self.cont_reads = _np(cont_reads, np.int32)
        self.cont_contig = _np(cont_contig, np.int32)
        nb, nc = int(self.read_len.sum()), self.cont_reads.size
        cap = int(feat_cap if feat_cap is not None else n * 8 + 1024)
        self.n, self.nb, self.nc, self.cap, self.lossy = n, nb, nc, cap, lossy_spec is not None
        self.f_op, self.f_pos = np.zeros(cap + 1, np.uint8), np.zeros(cap + 1, np.int32)
        self.f_a, self.f_b = np.zeros(cap + 1, np.int32), np.zeros(cap + 1, np.int32)
        self.f_off = np.zeros(n + 1, np.int64)
        self.matrices, self.hist = np.zeros(nc * 5, np.uint8), np.zeros(nc * 25, np.int64)
        self.scores, self.force = np.zeros(nb + 1, np.uint8), np.zeros(n + 1, np.uint8)
        self.status = np.zeros(n + 1, np.int32)
        hr = _lib.HostReads()
        hr.n_reads, hr.n_cont, hr.n_bases, hr.n_cigar_ops = n, nc, nb, int(self.n_cigar.sum())
        hr.bases, hr.quals = self.bases.ctypes.data, self.quals.ctypes.data if self.quals is not None else None
        hr.read_len, hr.aln_start = self.read_len.ctypes.data, self.aln_start.ctypes.data
        hr.rflags, hr.mapq = self.rflags.ctypes.data, self.mapq.ctypes.data
        hr.cigar, hr.n_cigar = self.cigar.ctypes.data, self.n_cigar.ctypes.data
        hr.cont_reads, hr.cont_contig = self.cont_reads.ctypes.data, self.cont_contig.ctypes.data
        out = _lib.HostReadResults()
        out.feats.op, out.feats.pos = self.f_op.ctypes.data, self.f_pos.ctypes.data
        out.feats.a, out.feats.b = self.f_a.ctypes.data, self.f_b.ctypes.data
        out.feats.off, out.feats.cap = self.f_off.ctypes.data, cap
        out.matrices, out.sub_hist = self.matrices.ctypes.data, self.hist.ctypes.data
        if self.lossy:
            out.scores, out.force, out.status = self.scores.ctypes.data, self.force.ctypes.data, self.status.ctypes.data
        self._hr, self._out = hr, out
        self.job = _lib.lib().cramgpu_submit_reads(engine.handle, C.byref(hr), lossy_spec.encode() if self.lossy else None,
                                                   int(window), C.byref(out))
        if not self.job:
            raise CramGpuError(_lib.E_ARG, _lib.last_error())

    def done(self):
        return _lib.check(_lib.lib().cramgpu_job_poll(self.job)) == 1

    def wait(self):
        """-> dict(op, pos, a, b, off, matrices[n_cont,5], hist[n_cont,25][, scores, force, status])."""
        rc = _lib.lib().cramgpu_job_wait(self.job)
        if rc == _lib.E_INSUFFICIENT:
            raise CramGpuError(rc, "feature capacity %d, needed %d" % (self.cap, int(self.f_off[self.n])))
        _lib.check(rc)
        k = int(self.f_off[self.n])
        res = dict(op=self.f_op[:k], pos=self.f_pos[:k], a=self.f_a[:k], b=self.f_b[:k], off=self.f_off,
                   matrices=self.matrices.reshape(self.nc, 5), hist=self.hist.reshape(self.nc, 25))
        if self.lossy:
            res.update(scores=self.scores[:self.nb], force=self.force[:self.n], status=self.status[:self.n])
        return res


class HostRestoreJob:
    """One outstanding cramgpu_submit_restore (CramNormalizer.restoreReadBases + restoreQualityScores)."""

    def __init__(self, engine, read_len, aln_start, rflags, cont_reads, cont_contig, feats, matrices, seq=None, seq_off=None,
                 scores=None, windows=None, default_qs=DEFAULT_QUALITY_SCORE):
        n = len(read_len)
        self.read_len, self.aln_start = _np(read_len, np.int32), _np(aln_start, np.int32)
        self.rflags = _np(rflags, np.uint8)
        self.cont_reads, self.cont_contig = _np(cont_reads, np.int32), _np(cont_contig, np.int32)
        self.f = [_np(feats["op"], np.uint8), _np(feats["pos"], np.int32), _np(feats["a"], np.int32),
                  _np(feats["b"], np.int32), _np(feats["off"], np.int64)]
        self.matrices = _np(matrices, np.uint8).reshape(-1)
        self.seq = _np(seq, np.uint8) if seq is not None else np.zeros(16, np.uint8)
        self.seq_off = _np(seq_off, np.int64) if seq_off is not None else None
        nb = int(self.read_len.sum())
        self.n, self.nb = n, nb
        self.bases = np.zeros(nb + 1, np.uint8)
        self.scores = _np(scores, np.uint8).copy() if scores is not None else None
        self.st_b, self.st_q = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
        hr = _lib.HostRecords()
        hr.n_reads, hr.n_cont, hr.n_seq = n, self.cont_reads.size, self.seq.size if seq is not None else 0
        hr.read_len, hr.aln_start, hr.rflags = self.read_len.ctypes.data, self.aln_start.ctypes.data, self.rflags.ctypes.data
        hr.cont_reads, hr.cont_contig = self.cont_reads.ctypes.data, self.cont_contig.ctypes.data
        if windows is not None:
            w = _np(windows, np.int64).reshape(-1, 2)
            self.w0, self.wl = _np(w[:, 0], np.int64), _np(w[:, 1], np.int64)
            hr.cont_win_start0, hr.cont_win_len = self.w0.ctypes.data, self.wl.ctypes.data
        hr.feats.op, hr.feats.pos, hr.feats.a, hr.feats.b, hr.feats.off = [x.ctypes.data for x in self.f]
        hr.feats.cap = self.f[0].size
        hr.seq, hr.seq_off = self.seq.ctypes.data, self.seq_off.ctypes.data if self.seq_off is not None else None
        hr.matrices, hr.default_qs = self.matrices.ctypes.data, int(default_qs)
        out = _lib.HostRestoreResults()
        out.bases, out.status_bases = self.bases.ctypes.data, self.st_b.ctypes.data
        if self.scores is not None:
            out.scores, out.status_quals = self.scores.ctypes.data, self.st_q.ctypes.data
        self._hr, self._out = hr, out
        self.job = _lib.lib().cramgpu_submit_restore(engine.handle, C.byref(hr), C.byref(out))
        if not self.job:
            raise CramGpuError(_lib.E_ARG, _lib.last_error())

    def wait(self):
        """-> (bases, status_bases, scores or None, status_quals or None)."""
        _lib.check(_lib.lib().cramgpu_job_wait(self.job))
        if self.scores is None:
            return self.bases[:self.nb], self.st_b[:self.n], None, None
        return self.bases[:self.nb], self.st_b[:self.n], self.scores[:self.nb], self.st_q[:self.n]
