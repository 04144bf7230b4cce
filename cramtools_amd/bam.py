"""Container assembly and BAM records on the device: the host-side mirror of what stands
between `CramNormalizer#normalize` and the BAM stripe in the reference's
src/main/java/htsjdk/samtools/cram/paralell/CramToBam_OBA_Function.java:64-75 — `Cram2SamRecordFactory#create`
(CIGAR from features), `BAMRecordCodec#encode` — and between the stripe of Bam_OBA_Supplier.java:45-82 and the columns
`Sam2CramRecordFactory` takes (`BAMRecordCodec#decode`).  Thin wrappers over the C ABI (include/cramgpu.h); there is
no CPU path here."""
import ctypes as C

import numpy as np

from . import _lib
from .records import FeatureBatch, RecordColumns, _dev, _stream, _torch  # noqa: F401


class Cram2SamRecordFactory:
    """htsjdk!build/Cram2SamRecordFactory: #getCigar2 (L119-225; the algorithm in source in the reference tree:
    encoding/reader/ReadFeatureBuffer.java:180-294) and the mate restoration that precedes it in
    CramNormalizer.java:82-105, :255-285, :455-470."""

    def __init__(self, engine, device):
        self.engine, self.device = engine, device

    def cigars(self, read_len, aln_start, bam_flags, feats, cigar_cap=None):
        """-> (n_cigar[n], cigar_off[n + 1], cigar uint32-as-int32, aln_end[n]) device tensors.  Synchronises once
        (status); raises CramGpuError(E_INSUFFICIENT) with the need in .args when cigar_cap is too small."""
        torch = _torch()
        n = int(read_len.numel())
        cap = int(cigar_cap) if cigar_cap is not None else n + 2 * int(feats.count) + 1
        n_cigar = torch.zeros(n, dtype=torch.int32, device=self.device)
        off = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        cigar = torch.zeros(max(cap, 1), dtype=torch.int32, device=self.device)
        aln_end = torch.zeros(n, dtype=torch.int32, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        fs = feats.struct()
        _lib.check(_lib.lib().cramgpu_cigars_dev(self.engine.handle, n, read_len.data_ptr(), aln_start.data_ptr(),
                                                 bam_flags.data_ptr(), C.byref(fs), n_cigar.data_ptr(), off.data_ptr(),
                                                 cigar.data_ptr(), cap, aln_end.data_ptr(), status.data_ptr(), _stream()))
        st = int(status.item())
        if st != 0:
            raise _lib.CramGpuError(st, "cigars: capacity %d, need %d" % (cap, int(off[n].item())))
        return n_cigar, off, cigar, aln_end

    def restore_mates(self, cols, cont_first, cont_ref_id, aln_start, aln_end):
        """CramNormalizer.java:82-105 for every container; cols is updated in place.  -> status[n_cont] (device)."""
        torch = _torch()
        n_cont = int(cont_first.numel()) - 1
        status = torch.zeros(n_cont, dtype=torch.int32, device=self.device)
        cs = cols.struct()
        self._keep = (cont_first, cont_ref_id)
        _lib.check(_lib.lib().cramgpu_restore_mates_dev(self.engine.handle, cols.n, n_cont, cont_first.data_ptr(), C.byref(cs),
                                                        cont_ref_id.data_ptr(), aln_start.data_ptr(), aln_end.data_ptr(),
                                                        status.data_ptr(), _stream()))
        return status


class CramSerilization:
    """src/main/java/htsjdk/samtools/cram/CramSerilization.java (the reference spells it so): what #convert does to a
    container's records between Sam2CramRecordFactory and ContainerFactory#buildContainer."""

    def __init__(self, engine, device):
        self.engine, self.device = engine, device

    def set_mate_info(self, cols, cont_first, aln_start, aln_end, names, name_off, name_len):
        """#setMateInfo (:274-373) for every container; cols.comp_flags / cols.next_frag are updated in place.
        names: uint8 device tensor the offsets point into (e.g. the BAM stripe), name_off int64[n], name_len int32[n]."""
        n_cont = int(cont_first.numel()) - 1
        cs = cols.struct()
        _lib.check(_lib.lib().cramgpu_set_mate_info_dev(self.engine.handle, cols.n, n_cont, cont_first.data_ptr(), C.byref(cs),
                                                        aln_start.data_ptr(), aln_end.data_ptr(), names.data_ptr(),
                                                        name_off.data_ptr(), name_len.data_ptr(), _stream()))

    def content_digests(self, cont_first, base_off, read_len, bases, quals):
        """ContentDigests BD / SD per container (cramgpu_content_digests_dev).  -> uint32-as-int64 tensor [n_cont, 2]."""
        torch = _torch()
        n_cont = int(cont_first.numel()) - 1
        out = torch.zeros(2 * n_cont, dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().cramgpu_content_digests_dev(self.engine.handle, int(read_len.numel()), n_cont, cont_first.data_ptr(),
                                                          base_off.data_ptr(), read_len.data_ptr(), bases.data_ptr(),
                                                          quals.data_ptr(), out.data_ptr(), _stream()))
        return (out.to(torch.int64) & 0xFFFFFFFF).reshape(n_cont, 2)


# MessageDigestHasher#asByteArray hands out messageDigest.digest() of the digest #getValue has just reset: the digest
# of the empty message (SHA-512 / SHA-1 of b""), whatever the records held (see include/cramgpu.h).
_SHA512_EMPTY = bytes.fromhex("cf83e1357eefb8bdf1542850d66d8007d620e4050b5715dc83f4a921d36ce9ce"
                              "47d0d13c5d85f2b0ff8318d2877eec2f63b931bd47417a81a538327af927da3e")
_SHA1_EMPTY = bytes.fromhex("da39a3ee5e6b4b0d3255bfef95601890afd80709")


def slice_content_tags(bd, sd):
    """The six tags ContentDigests.create(ALL).getAsTags() gives a slice, as (tag, byte-array value) in the order of
    KNOWN_DIGESTS: BD / SD = the CRC sums little-endian (Crc32Hasher#asByteArray), B5 S5 B1 S1 as described above."""
    le = lambda v: int(v & 0xFFFFFFFF).to_bytes(4, "little")
    return [("BD", le(bd)), ("SD", le(sd)), ("B5", _SHA512_EMPTY), ("S5", _SHA512_EMPTY), ("B1", _SHA1_EMPTY), ("S1", _SHA1_EMPTY)]


class BAMRecordCodec:
    """htsjdk!BAMRecordCodec over whole stripes: encode() = one BAM alignment record per record, back to back (layout:
    encoding/reader/BAMRecordView.java:23-35 ...), decode() = the columns of a stripe of records."""

    def __init__(self, engine, device):
        self.engine, self.device = engine, device

    def encode(self, cols, read_len, aln_start, aln_end, mapq, base_off, bases, quals, n_cigar, cigar_off, cigar, bam_cap,
               qual_status=None, names=None, name_off=None, name_base=0, tags=None, tag_off=None):
        """cols: RecordColumns after restore_mates (ref_id per record).  -> (bam uint8[bam_cap], rec_off[n + 1],
        status[n]) device tensors; stream-ordered."""
        torch = _torch()
        n = int(read_len.numel())
        b = _lib.BamCols()
        b.ref_id, b.aln_start, b.aln_end, b.mapq = cols.ref_id.data_ptr(), aln_start.data_ptr(), aln_end.data_ptr(), mapq.data_ptr()
        b.bam_flags, b.mate_flags, b.mate_ref = cols.bam_flags.data_ptr(), cols.mate_flags.data_ptr(), cols.mate_ref.data_ptr()
        b.mate_start, b.tlen, b.read_len = cols.mate_start.data_ptr(), cols.tlen.data_ptr(), read_len.data_ptr()
        b.base_off, b.bases, b.quals = base_off.data_ptr(), bases.data_ptr(), quals.data_ptr()
        b.qual_status = qual_status.data_ptr() if qual_status is not None else None
        b.n_cigar, b.cigar_off, b.cigar = n_cigar.data_ptr(), cigar_off.data_ptr(), cigar.data_ptr()
        b.names = names.data_ptr() if names is not None else None
        b.name_off = name_off.data_ptr() if name_off is not None else None
        b.name_base = int(name_base)
        b.tags = tags.data_ptr() if tags is not None else None
        b.tag_off = tag_off.data_ptr() if tag_off is not None else None
        bam = torch.zeros(int(bam_cap) + 16, dtype=torch.uint8, device=self.device)
        rec_off = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        status = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().cramgpu_bam_encode_dev(self.engine.handle, n, C.byref(b), bam.data_ptr(), int(bam_cap),
                                                     rec_off.data_ptr(), status.data_ptr(), _stream()))
        return bam, rec_off, status[:n]

    @staticmethod
    def index(stripe):
        """Record offsets of a host stripe (numpy uint8): Bam_OBA_Supplier.java:49-62.  -> int64[n + 1]."""
        stripe = np.ascontiguousarray(stripe, np.uint8)
        n = _lib.check(_lib.lib().cramgpu_bam_index(stripe.ctypes.data, stripe.size, None, 0))
        off = np.zeros(n + 1, np.int64)
        _lib.check(_lib.lib().cramgpu_bam_index(stripe.ctypes.data, stripe.size, off.ctypes.data, n + 1))
        return off

    def index_dev(self, bam, length, n_ref, cap):
        """Record offsets of a stripe in device memory (cramgpu_bam_index_dev).  -> (rec_off int64[cap] device tensor, number
        of records, status); synchronises to read the two scalars."""
        torch = _torch()
        rec_off = torch.zeros(max(int(cap), 1), dtype=torch.int64, device=self.device)
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().cramgpu_bam_index_dev(self.engine.handle, bam.data_ptr(), int(length), int(n_ref), rec_off.data_ptr(),
                                                    int(cap), count.data_ptr(), status.data_ptr(), _stream()))
        return rec_off, int(count.item()), int(status.item())

    def decode(self, bam, rec_off, bases_cap, cigar_cap):
        """bam / rec_off: device tensors.  -> dict of device tensors (the cramgpu_bam_out columns) + 'status'."""
        torch = _torch()
        n = int(rec_off.numel()) - 1
        z = lambda k, dt: torch.zeros(k, dtype=dt, device=self.device)
        o = dict(ref_id=z(n, torch.int32), aln_start=z(n, torch.int32), mapq=z(n, torch.uint8), bam_flags=z(n, torch.int32),
                 mate_ref=z(n, torch.int32), mate_start=z(n, torch.int32), tlen=z(n, torch.int32), read_len=z(n, torch.int32),
                 base_off=z(n + 1, torch.int64), bases=z(int(bases_cap) + 16, torch.uint8), quals=z(int(bases_cap) + 16, torch.uint8),
                 n_cigar=z(n, torch.int32), cigar_off=z(n + 1, torch.int64), cigar=z(max(int(cigar_cap), 1), torch.int32),
                 aln_span=z(n, torch.int32), rflags=z(n, torch.uint8), name_off=z(n, torch.int64), name_len=z(n, torch.int32),
                 tag_off=z(n, torch.int64), tag_len=z(n, torch.int32))
        s = _lib.BamOut()
        for k, v in o.items():
            setattr(s, k, v.data_ptr())
        s.bases_cap, s.cigar_cap = int(bases_cap), int(cigar_cap)
        status = z(max(n, 1), torch.int32)
        _lib.check(_lib.lib().cramgpu_bam_decode_dev(self.engine.handle, n, bam.data_ptr(), rec_off.data_ptr(), C.byref(s),
                                                     status.data_ptr(), _stream()))
        o["status"] = status[:n]
        return o


def slice_ref_md5(engine, device, d_ref, windows, start, span):
    """htsjdk!structure/Slice#setRefMD5 for a batch of slices (cramgpu_slice_md5_dev): windows = (ref_off, ref_len,
    ref_start0) per slice, start / span = the slices' 1-based alignment start and span.  -> uint8[n, 16] device tensor."""
    torch = _torch()
    n = len(start)
    win = _dev(np.asarray(windows, np.int64).reshape(-1), torch.int64, device)
    st, sp = _dev(start, torch.int32, device), _dev(span, torch.int32, device)
    out = torch.zeros(max(n, 1) * 16, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().cramgpu_slice_md5_dev(engine.handle, n, d_ref.data_ptr(), win.data_ptr(), st.data_ptr(), sp.data_ptr(),
                                                out.data_ptr(), _stream()))
    return out[:n * 16].reshape(n, 16)

