#!/usr/bin/env python
"""bench.py — CRAM encode hot path on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the encode hot path over the whole synthetic read set resident in HBM:
  K3  reference-diff feature extraction over every read (two passes + scan, substitution histogram),
  F3  substitution matrices + code assignment (device, per container),
  W   data-series writer: every record's values into one byte stream per series per container (the K1 input arena),
  K1  rANS order-0 / order-1 encode of every external block of every container (QS, BA, BF, CF, NS, RG, RL, TS
      order 1; AP, RI order 0 — the fixed series->codec map of CompressionHeaderFactory#build L82-124),
  gather of the compressed streams into one dense arena (what would cross PCIe).
Workload: 10M x 150 bp Illumina-like reads vs a 64 Mbp synthetic reference, 1000 containers of 10 000 reads,
lossless qualities.  The external-block byte streams are made inside the step by the device data-series writer (W,
Writer#write of SURVEY.md §8(f) row 1) from the record columns and K3's features; the same streams built with torch at
set-up time stay as the expected corpus (decode leg, CPU baseline, parity).

Contract: python bench.py --gpus N --steps K --warmup W ; N > 1 under torch.distributed.run, one rank per GPU, each
rank codes its own shard of the same size (weak scaling; containers are independent, no collective on the data path).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Hardware queues of this process (read by the HIP runtime at its first call): launches of streams that share a queue
# serialise, and the decode leg (K2 beside gunzip), the job slots and the coalescing queue overlap launches of several
# streams.  HIP's default of 4 leaves three for them (scripts/ubench/streams_concurrency.py); the bench asks for 8,
# as INTEGRATION.md tells a JVM host to, and reports it (`hw_queues`).  BENCH_HW_QUEUES overrides.
HW_QUEUES = max(1, min(32, int(os.environ.get("BENCH_HW_QUEUES", "8"))))
os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)

import numpy as np
import torch

from cramtools_amd import _lib, sharding, synth_torch
from cramtools_amd import external_compression as xc
from cramtools_amd import records as rec

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
READS_PER_CONTAINER = 10000  # CramContext.java:33
# series of the block corpus: (name, rANS order) — CompressionHeaderFactory#build L82-124
BLOCK_DTYPE = np.dtype([("in_off", np.int64), ("out_off", np.int64), ("in_len", np.int32), ("out_cap", np.int32),
                        ("order", np.int32), ("reserved", np.int32)])  # cramgpu_block as a numpy record (tables built without a Python loop)
SERIES = [("QS", 1), ("AP", 0), ("RI", 0), ("BA", 1), ("BF", 1), ("CF", 1), ("NS", 1), ("RG", 1), ("RL", 1), ("TS", 1)]


def gpu_insert_bases(eng, reads, ref, n_reads):
    """(is-InsertBase mask, payload, per-read feature offsets) of one K3 run — used to lay out the BA series."""
    fac = rec.Sam2CramRecordFactory(eng)
    fac.set_ref_bases(ref)
    feats, _ = fac.create_variations(reads, feature_cap=n_reads * 2 + 1024, with_hist=False)
    n = feats.count
    return feats.op[:n].clone(), feats.a[:n].clone(), feats.off.clone()


def _concat_read_sets(parts, dev):
    """Read sets of consecutive contigs as one (offsets shifted; the 16-byte / 1-element tails of each part dropped)."""
    out = synth_torch.ReadSet()
    out.n = sum(p.n for p in parts)
    b_off = c_off = 0
    base_off, cigar_off, bases, quals, cigar = [], [], [], [], []
    for p in parts:
        nb, ncg = p.bases.numel() - 16, p.cigar.numel() - 1
        base_off.append(p.base_off + b_off)
        cigar_off.append(p.cigar_off + c_off)
        bases.append(p.bases[:nb])
        quals.append(p.quals[:nb])
        cigar.append(p.cigar[:ncg])
        b_off += nb
        c_off += ncg
    pad = torch.zeros(16, dtype=torch.uint8, device=dev)
    out.bases, out.quals = torch.cat(bases + [pad]), torch.cat(quals + [pad])
    out.cigar = torch.cat(cigar + [torch.zeros(1, dtype=torch.int32, device=dev)])
    out.base_off, out.cigar_off = torch.cat(base_off), torch.cat(cigar_off)
    for name in ("read_len", "aln_start", "n_cigar", "mapq", "rflags", "aln_span", "flags_rnd"):
        setattr(out, name, torch.cat([getattr(p, name) for p in parts]))
    return out


def build_workload(dev, n_reads, ref_len, read_len, seed_shift, features_of, n_contigs=1):
    """Reads + reference + the per-container external-block corpus, all on `dev`.
    features_of(reads, ref) -> (op, a, off) of the reads' features (the GPU kernel in the bench, the oracle in tests).
    n_contigs > 1 (the per-GPU shard shape of configs[3]): the reference is n_contigs contigs of ref_len / n_contigs
    bases back to back in one arena (16-byte aligned starts), every contig gets the same share of the reads (a whole
    number of containers: CRAMContainerStreamWriter flushes on a reference change), alignment starts are contig-local
    and every container's window names its contig (non-zero ref_off)."""
    if n_contigs == 1:
        ref = synth_torch.reference(ref_len, device=dev)
        ref = torch.cat([ref, torch.zeros(16, dtype=torch.uint8, device=dev)])
        rs = synth_torch.make_reads(n_reads, ref[:ref_len], read_len=read_len, seed=1 + seed_shift, qual_seed=2 + seed_shift,
                                    err_seed=3 + seed_shift)
        n_cont = (n_reads + READS_PER_CONTAINER - 1) // READS_PER_CONTAINER
        windows = [(0, ref_len, 0)] * n_cont
    else:
        per = n_reads // n_contigs
        assert per * n_contigs == n_reads and per % READS_PER_CONTAINER == 0, "whole containers per contig"
        clen = ref_len // n_contigs
        stride = (clen + 15) // 16 * 16 + 16
        whole = synth_torch.reference(clen * n_contigs, device=dev)
        ref = torch.zeros(stride * n_contigs + 16, dtype=torch.uint8, device=dev)
        parts, windows = [], []
        for k in range(n_contigs):
            ref[k * stride:k * stride + clen] = whole[k * clen:(k + 1) * clen]
            parts.append(synth_torch.make_reads(per, ref[k * stride:k * stride + clen], read_len=read_len,
                                                seed=1 + seed_shift + 7 * k, qual_seed=2 + seed_shift + 7 * k,
                                                err_seed=3 + seed_shift + 7 * k))
            windows += [(k * stride, clen, 0)] * (per // READS_PER_CONTAINER)
        del whole
        rs = _concat_read_sets(parts, dev)
        del parts
        n_cont = n_reads // READS_PER_CONTAINER
    read_cont = (torch.arange(n_reads, device=dev) // READS_PER_CONTAINER).to(torch.int32)
    reads = rec.ReadBatch(dev, rs.bases, rs.base_off, rs.read_len, rs.aln_start, rs.cigar, rs.cigar_off, rs.n_cigar,
                          quals=rs.quals, mapq=rs.mapq, rflags=rs.rflags, read_cont=read_cont, windows=windows)
    # ---- value series of the records (synthetic but shaped like the reference's) ----
    fr = rs.flags_rnd
    cont_first = torch.arange(n_cont, device=dev) * READS_PER_CONTAINER
    starts = rs.aln_start.to(torch.int64)
    ap = starts.clone()
    ap[1:] = starts[1:] - starts[:-1]
    ap[cont_first] = starts[cont_first] - starts[cont_first]  # AP delta restarts at 0 in every container (slice start)
    bf = torch.tensor([99, 147, 83, 163], device=dev)[(fr >> 9) % 4]  # paired, proper, strand/mate bits
    cf = torch.full((n_reads,), 3, device=dev)  # force-preserve QS | detached
    ri = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    ns = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    rg = torch.full((n_reads,), -1, device=dev)
    rl = rs.read_len.to(torch.int64)
    ts = (300 + ((fr >> 11) % 101) - 50) * torch.where(((fr >> 9) & 1) == 0, 1, -1)
    # the same values as the columns of the records (what Writer#write reads): the device writer makes the streams from
    # these inside the timed step; the torch-made streams below stay as the expected corpus (decode leg, CPU baseline)
    cols = rec.RecordColumns(dev, n_reads, multiref=torch.ones(n_cont, dtype=torch.uint8, device=dev), ap_delta=True, alias_qs=True,
                             bam_flags=bf, comp_flags=cf, ref_id=ri, read_group=rg, mate_flags=(bf >> 5) & 1, mate_ref=ns,
                             mate_start=torch.clamp(starts + ts, min=1), tlen=ts)
    series_bytes = {}
    for name, vals in (("AP", ap), ("RI", ri), ("BF", bf), ("CF", cf), ("NS", ns), ("RG", rg), ("RL", rl), ("TS", ts)):
        b, off = synth_torch.itf8(vals)
        series_bytes[name] = (b, off[cont_first], torch.cat([off[cont_first[1:]], off[-1:]]))
    # QS: with lossless qualities every record carries its quality array (Writer#write L139-261)
    q_lo = rs.base_off[cont_first]
    q_hi = torch.cat([q_lo[1:], torch.tensor([n_reads * read_len], device=dev)])
    series_bytes["QS"] = (rs.quals, q_lo, q_hi)
    # BA: bases of InsertBase features, in record order (one K3 run at set-up time)
    f_op, f_a, f_off = features_of(reads, ref)
    n_feat = int(f_op.numel())
    is_i = f_op == ord("i")
    ba = f_a[is_i].to(torch.uint8)
    ci = torch.zeros(n_feat + 1, dtype=torch.int64, device=dev)
    ci[1:] = torch.cumsum(is_i.to(torch.int64), 0)
    f_lo = f_off[cont_first]
    f_hi = torch.cat([f_off[cont_first[1:]], f_off[-1:]])
    series_bytes["BA"] = (torch.cat([ba, torch.zeros(16, dtype=torch.uint8, device=dev)]), ci[f_lo], ci[f_hi])
    feat_cap = n_feat + 1024
    del f_op, f_a, f_off
    # ---- pack every (container, series) piece at 16-byte aligned offsets of one input arena ----
    lens = torch.stack([series_bytes[name][2] - series_bytes[name][1] for name, _ in SERIES], 1)  # [n_cont, n_series]
    lens_h = lens.cpu().numpy().astype(np.int64)
    orders = np.tile(np.array([o for _, o in SERIES], np.int32), n_cont)
    flat_len = lens_h.reshape(-1)
    blocks, in_bytes, out_bytes = xc.Engine.plan_encode(flat_len, orders)
    n_blocks = flat_len.size
    d_in = torch.zeros(in_bytes, dtype=torch.uint8, device=dev)
    in_off = np.array([blocks[i].in_off for i in range(n_blocks)], np.int64).reshape(n_cont, len(SERIES))
    for k, (name, _) in enumerate(SERIES):
        src, lo, hi = series_bytes[name]
        total = int((hi - lo).sum().item())
        if total == 0:
            continue
        # piece c copies src[lo[c]:hi[c]] to d_in[in_off[c,k]:...]: one flat index gather per series
        ln = (hi - lo)
        if total > (256 << 20):  # large pieces (QS): one plain copy per container
            lo_h, hi_h = lo.cpu().numpy(), hi.cpu().numpy()
            for c in range(n_cont):
                d_in[in_off[c, k]:in_off[c, k] + int(hi_h[c] - lo_h[c])] = src[int(lo_h[c]):int(hi_h[c])]
            continue
        dst0 = torch.as_tensor(in_off[:, k], device=dev)
        rep = torch.repeat_interleave(torch.arange(n_cont, device=dev), ln)
        within = torch.arange(total, device=dev) - torch.repeat_interleave(torch.cumsum(ln, 0) - ln, ln)
        d_in[dst0[rep] + within] = src[lo[rep] + within]
        del rep, within
    return dict(ref=ref, reads=reads, rs=rs, cols=cols, n_cont=n_cont, blocks=blocks, n_blocks=n_blocks, d_in=d_in,
                out_bytes=out_bytes, raw_bytes=int(flat_len.sum()), feat_cap=feat_cap, lens=lens_h, max_cap=max(
                    blocks[i].out_cap for i in range(n_blocks)))


class Step:
    """Pre-allocated device state of one encode step (nothing is allocated inside the timed region)."""

    def __init__(self, dev, eng, w, writer=True):
        self.dev, self.eng, self.w = dev, eng, w
        n = w["n_blocks"]
        # W: the data-series writer produces the K1 input arena inside the step (all 30 series of every container; the
        # ten rANS series are the blocks, the others are what the host would gzip)
        self.writer = rec.Writer(eng, dev, w["n_cont"], w["d_in"].numel() + 40 * w["reads"].n + 16 * w["feat_cap"] +
                                 512 * w["n_cont"]) if writer else None
        self.ids = np.array([_lib.SERIES_ID[name] for name, _ in SERIES])
        self.orders = np.array([o for _, o in SERIES], np.int32)
        self.blk = np.zeros(n, BLOCK_DTYPE)
        self.blk["order"] = np.tile(self.orders, w["n_cont"])
        self.d_blk = torch.zeros(32 * n, dtype=torch.uint8, device=dev)  # the block table cramgpu_rans_encode_series_dev makes
        self.d_blk_status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_out = torch.empty(w["out_bytes"], dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(n, dtype=torch.int32, device=dev)
        self.d_pos = torch.zeros(n, dtype=torch.int64, device=dev)
        self.d_pack = torch.empty(w["raw_bytes"] + 32 * n + (64 << 20), dtype=torch.uint8, device=dev)
        self.d_pack_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        self.feats = rec.FeatureBatch(dev, w["reads"].n, w["feat_cap"])
        self.hist = torch.zeros(w["n_cont"] * 25, dtype=torch.int64, device=dev)
        self.mats = torch.zeros(w["n_cont"] * 5, dtype=torch.uint8, device=dev)
        self.fac = rec.Sam2CramRecordFactory(eng)
        self.fac.set_ref_bases(w["ref"])
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.enc_ms = []

    def run(self, timed):
        w, L = self.w, _lib.lib()
        reads = w["reads"]
        s = torch.cuda.current_stream().cuda_stream
        # K3 + F3
        self.hist.zero_()
        rs_, fs_ = reads.struct(), self.feats.struct()
        n = _lib.check(L.cramgpu_extract_features_dev(self.eng.handle, C.byref(rs_), w["ref"].data_ptr(), C.byref(fs_),
                                                      self.hist.data_ptr(), s))
        self.feats.count = int(n)
        self.fac.matrices_and_codes(reads, self.feats, self.hist, self.mats)  # F3 on the device, stream-ordered
        d_in, blocks = w["d_in"], w["blocks"]
        if self.writer is not None:
            # W: records + features -> the byte streams of every series, then K1 straight behind it: the block table
            # (stream of series k of container c, its slot of RANS#allocateIfNeeded's size) and its longest-first order
            # are made on the device (cramgpu_rans_encode_series_dev), so nothing comes back to the host between the two
            # (the lengths went to the host and the table back up before: ~1 ms of an idle GPU per step).
            # Aliased QS streams lie in reads.quals (CRAMGPU_WS_ALIAS_QS): all offsets rebased on the lower address.
            self.writer.write(reads, w["cols"], self.feats)
            p_arena = self.writer.arena.data_ptr()
            base = min(p_arena, reads.quals.data_ptr()) & ~15
            if timed:
                self.ev[0].record()
            self.eng.encode_series_dev(base, p_arena - base, self.writer.off, self.writer.len, self.writer.status, w["n_cont"],
                                       self.ids, self.orders, self.d_out, self.d_blk, self.d_len, self.d_pos, self.d_blk_status)
        else:
            # K1 over the workload's own arena (round-1 form)
            if timed:
                self.ev[0].record()
            self.eng.encode_dev(d_in, self.d_out, blocks, w["n_blocks"], self.d_len, self.d_pos)
        if timed:
            self.ev[1].record()
        self.eng.gather_dev(self.d_out, self.d_pos, self.d_len, w["n_blocks"], w["max_cap"], self.d_pack, self.d_pack_off)
        if timed:
            torch.cuda.synchronize()
            self.enc_ms.append(self.ev[0].elapsed_time(self.ev[1]))


    # ---- the whole container, as CramSerilization.java:108-200 does it, in ONE timed region (missing 2) ----
    def prepare_full(self):
        """Set-up of run_full (untimed): read names (ten decimal digits of the read's index: every name occurs once, so
        #setMateInfo must find every record detached, which is what the corpus' CF column says), alignment ends and slice
        spans from the CIGARs, output buffers."""
        w, dev, reads = self.w, self.dev, self.w["reads"]
        n, n_cont = reads.n, w["n_cont"]
        idx = torch.arange(n, device=dev, dtype=torch.int64)
        self.names = ((idx[:, None] // (10 ** torch.arange(9, -1, -1, device=dev, dtype=torch.int64))[None, :]) % 10 + 48).to(torch.uint8).reshape(-1).contiguous()
        self.name_off, self.name_len = (idx * 10).contiguous(), torch.full((n,), 10, dtype=torch.int32, device=dev)
        op, ln = (reads.cigar & 15).to(torch.int64), (reads.cigar >> 4).to(torch.int64)
        owner = torch.repeat_interleave(torch.arange(n, device=dev), reads.n_cigar.to(torch.int64))
        span = torch.zeros(n, dtype=torch.int64, device=dev)
        on_ref = (op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8)
        span.index_add_(0, owner, torch.where(on_ref, ln, torch.zeros_like(ln))[:owner.numel()])
        self.aln_end = (reads.aln_start.to(torch.int64) + span - 1).to(torch.int32)
        self.cont_first = torch.as_tensor(np.minimum(np.arange(n_cont + 1, dtype=np.int64) * READS_PER_CONTAINER, n).astype(np.int32)).to(dev)
        cont = reads.read_cont.to(torch.int64)
        hi = torch.zeros(n_cont, dtype=torch.int64, device=dev).scatter_reduce_(0, cont, self.aln_end.to(torch.int64), "amax")
        lo = reads.aln_start[self.cont_first[:-1].to(torch.int64)].to(torch.int64)
        self.sl_start, self.sl_span = lo.to(torch.int32), (hi - lo + 1).to(torch.int32)
        self.sl_win = reads.win  # (ref_off, ref_len, ref_start0) per container: the whole contig on encode
        self.md5 = torch.zeros(n_cont * 16, dtype=torch.uint8, device=dev)
        self.crc = torch.zeros(n_cont * 2, dtype=torch.int32, device=dev)
        self.rest = np.array([k for k in range(_lib.N_SERIES) if k not in set(self.ids.tolist())])
        self.gz_piece = 8192
        self.gz_desc = None
        self.side = torch.cuda.Stream(device=dev)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.ev_w = torch.cuda.Event()
        self.pin_off = torch.zeros(n_cont * _lib.N_SERIES, dtype=torch.int64).pin_memory()
        self.pin_len = torch.zeros(n_cont * _lib.N_SERIES, dtype=torch.int32).pin_memory()
        self.pin_st = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.ev_full = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.full_ms = []
        from cramtools_amd import bam as bam_mod
        self.ser = bam_mod.CramSerilization(self.eng, dev)

    def _gzip_plan(self, off, ln):
        """Descriptors of the GZIP series (every series the fixed map does not give to rANS), longer ones cut into pieces
        of gz_piece bytes, one member each, longest first; numpy only (inside the timed region)."""
        o, l = off[:, self.rest].reshape(-1), ln[:, self.rest].reshape(-1)
        keep = l > 0
        o, l = o[keep], l[keep]
        k = (l + self.gz_piece - 1) // self.gz_piece
        src = np.repeat(o, k)
        within = np.arange(int(k.sum()), dtype=np.int64) - np.repeat(np.cumsum(k) - k, k)
        start = within * self.gz_piece
        pl = np.minimum(self.gz_piece, np.repeat(l, k) - start)
        d = np.zeros(pl.size, BLOCK_DTYPE)
        d["in_off"], d["in_len"] = src + start, pl
        cap = (pl + 5 * ((pl >> 15) + 1) + 64 + 15) // 16 * 16  # >= cramgpu_gzip_bound: input + 5 per stored block + header / trailer
        d["out_cap"] = cap
        d["out_off"] = np.cumsum(cap) - cap
        return d[np.argsort(-pl, kind="stable")], int(cap.sum())

    def run_full(self, timed):
        """K3 -> F3 -> #setMateInfo -> W -> Z (the GZIP series as gzip members) -> K1 -> gather, with the slice reference MD5s
        and the ContentDigests on a second stream beside all of it; one region."""
        w, L = self.w, _lib.lib()
        reads, cols = w["reads"], w["cols"]
        main = torch.cuda.current_stream()
        if timed:
            self.ev_full[0].record()
        # slice MD5 and content digests read the reads and the reference only and use no LDS: on a second stream from the
        # head of the region, they fill the issue slots the other kernels (at last the encoder's chains) leave
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            sd = self.side.cuda_stream
            _lib.check(L.cramgpu_slice_md5_dev(self.eng.handle, w["n_cont"], w["ref"].data_ptr(), self.sl_win.data_ptr(),
                                               self.sl_start.data_ptr(), self.sl_span.data_ptr(), self.md5.data_ptr(), sd))
            _lib.check(L.cramgpu_content_digests_dev(self.eng.handle, reads.n, w["n_cont"], self.cont_first.data_ptr(),
                                                     reads.base_off.data_ptr(), reads.read_len.data_ptr(), reads.bases.data_ptr(),
                                                     reads.quals.data_ptr(), self.crc.data_ptr(), sd))
        self.hist.zero_()
        rs_, fs_ = reads.struct(), self.feats.struct()
        self.feats.count = int(_lib.check(L.cramgpu_extract_features_dev(self.eng.handle, C.byref(rs_), w["ref"].data_ptr(), C.byref(fs_),
                                                                         self.hist.data_ptr(), main.cuda_stream)))
        self.fac.matrices_and_codes(reads, self.feats, self.hist, self.mats)
        self.ser.set_mate_info(cols, self.cont_first, reads.aln_start, self.aln_end, self.names, self.name_off, self.name_len)
        self.writer.write(reads, cols, self.feats)
        self.ev_w.record(main)
        # K1 straight behind W (block table and launch order made on the device: cramgpu_rans_encode_series_dev), the
        # gather behind it.  The stream lengths go to the host on a stream of their own that waits for W only, and the
        # host lays out the gzip members while the encoder runs.
        p_arena = self.writer.arena.data_ptr()
        base = min(p_arena, reads.quals.data_ptr()) & ~15
        self.eng.encode_series_dev(base, p_arena - base, self.writer.off, self.writer.len, self.writer.status, w["n_cont"],
                                   self.ids, self.orders, self.d_out, self.d_blk, self.d_len, self.d_pos, self.d_blk_status)
        self.eng.gather_dev(self.d_out, self.d_pos, self.d_len, w["n_blocks"], w["max_cap"], self.d_pack, self.d_pack_off)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.ev_w)
            self.pin_off.copy_(self.writer.off, non_blocking=True)
            self.pin_len.copy_(self.writer.len, non_blocking=True)
            self.pin_st.copy_(self.writer.status, non_blocking=True)
        self.copy_stream.synchronize()
        if int(self.pin_st[0]) != 0:
            raise _lib.CramGpuError(int(self.pin_st[0]), "data-series writer")
        off = self.pin_off.numpy().reshape(w["n_cont"], _lib.N_SERIES)
        ln = self.pin_len.numpy().reshape(w["n_cont"], _lib.N_SERIES)
        gd, gz_bytes = self._gzip_plan(off, ln)
        if self.gz_desc is None:  # first (warm-up) call: buffers with headroom for later calls
            self.gz_desc = torch.zeros(int(gd.size * 1.25) + 1024, BLOCK_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
            self.gz_out = torch.zeros(int(gz_bytes * 1.25) + (1 << 20), dtype=torch.uint8, device=self.dev)
            self.gz_len = torch.zeros(self.gz_desc.shape[0], dtype=torch.int32, device=self.dev)
            self.gz_pin = torch.zeros(self.gz_desc.shape[0], BLOCK_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
        assert gd.size <= self.gz_desc.shape[0] and gz_bytes + 16 <= self.gz_out.numel()
        self.gz_pin[:gd.size] = torch.from_numpy(gd.view(np.uint8).reshape(gd.size, -1))
        self.gz_desc[:gd.size].copy_(self.gz_pin[:gd.size], non_blocking=True)
        self.n_gz, self.gz_plan = int(gd.size), gd
        # Z behind K1 on the same stream: the encoder's resident workgroups hold the whole LDS of every CU, a gzip wave
        # (9.5 KiB of tables) finds no room beside them and would only start as they retire
        self.eng.gzip_dev(self.writer.arena, self.gz_out, self.gz_desc, self.n_gz, self.gz_len)
        main.wait_stream(self.side)
        if timed:
            self.ev_full[1].record()
            torch.cuda.synchronize()
            self.full_ms.append(self.ev_full[0].elapsed_time(self.ev_full[1]))

    def verify_full(self):
        """What run_full left: every record detached (names are unique), slice MD5s and content digests of sampled
        containers against hashlib / zlib, every gzip member inflated by the gunzip kernel back to its series piece."""
        import hashlib
        import zlib
        w, reads, cols = self.w, self.w["reads"], self.w["cols"]
        assert bool(((cols.comp_flags & 2) != 0).all().item()) and bool((cols.next_frag == -1).all().item()), "setMateInfo: a record is not detached"
        ref_h = None
        for c in (0, w["n_cont"] // 2, w["n_cont"] - 1):
            lo, span = int(self.sl_start[c].item()), int(self.sl_span[c].item())
            seg = w["ref"][lo - 1:lo - 1 + span].cpu().numpy().tobytes()
            assert self.md5[16 * c:16 * c + 16].cpu().numpy().tobytes() == hashlib.md5(seg).digest(), "slice MD5 %d" % c
            r0, r1 = int(self.cont_first[c].item()), int(self.cont_first[c + 1].item())
            b0, b1 = int(reads.base_off[r0].item()), int(reads.base_off[r1 - 1].item()) + int(reads.read_len[r1 - 1].item())
            bases, quals = reads.bases[b0:b1].cpu().numpy(), reads.quals[b0:b1].cpu().numpy()
            offs = (reads.base_off[r0:r1] - b0).cpu().numpy()
            lens = reads.read_len[r0:r1].cpu().numpy()
            bd = sum(zlib.crc32(bases[o:o + l].tobytes()) for o, l in zip(offs, lens)) & 0xFFFFFFFF
            sd = sum(zlib.crc32(quals[o:o + l].tobytes()) for o, l in zip(offs, lens)) & 0xFFFFFFFF
            got = self.crc[2 * c:2 * c + 2].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            assert (int(got[0]), int(got[1])) == (bd, sd), "content digests %d" % c
        n, gd = self.n_gz, self.gz_plan
        zl = self.gz_len[:n].cpu().numpy()
        assert (zl > 0).all(), "a GZIP series failed to compress"
        back = np.zeros(n, BLOCK_DTYPE)
        back["in_off"], back["in_len"], back["out_cap"] = gd["out_off"], zl, gd["in_len"]
        pad = (gd["in_len"].astype(np.int64) + 31) // 16 * 16
        back["out_off"] = np.cumsum(pad) - pad
        d_back = torch.zeros(int(pad.sum()) + 16, dtype=torch.uint8, device=self.dev)
        d_blen = torch.zeros(n, dtype=torch.int32, device=self.dev)
        self.eng.gunzip_dev(self.gz_out, d_back, torch.as_tensor(back.view(np.uint8).copy()).to(self.dev), n, d_blen)
        torch.cuda.synchronize()
        assert np.array_equal(d_blen.cpu().numpy(), gd["in_len"]), "a gzip member does not inflate to its length"
        for i in list(range(0, n, max(1, n // 64))) + [n - 1]:  # the bytes of a sample of members
            a0, b0_, l = int(gd["in_off"][i]), int(back["out_off"][i]), int(gd["in_len"][i])
            assert torch.equal(d_back[b0_:b0_ + l], self.writer.arena[a0:a0 + l]), "gzip member %d" % i
        return dict(gzip_members=n, gzip_raw_bytes=int(gd["in_len"].sum()), gzip_bytes=int(zl.sum()))


class DecodeStep:
    """configs[2]: the decode hot path over the streams one encode step produced (nothing allocated while timed):
      K2  rANS decode of every external block (from the gathered dense arena back into a raw arena),
      G   gunzip of the series the reference stores as GZIP blocks (ExternalCompression#gunzip),
      R   the data-series reader (CramRecordReader#read): the decoded rANS series, and the series the reference gzips
          from where G inflates them, back into record columns, features, I / S
          bytes and quality arrays,
      N1  CramNormalizer.restoreReadBases for every read from the READER's columns and features (whole-reference window
          per container, matrices on the host as the compression header holds them),
      N2  CramNormalizer.restoreQualityScores on the reader's quality arrays,
      A   container assembly: CIGARs from the features (Cram2SamRecordFactory#getCigar2), mate
          restoration (CramNormalizer.java:82-105), one BAM record per read (BAMRecordCodec#encode) — the uncompressed
          stripe CramToBam_OBA_Function hands to BGZF."""

    def __init__(self, dev, eng, w, enc):
        self.dev, self.eng, self.w, self.enc = dev, eng, w, enc
        n = w["n_blocks"]
        lens = enc.d_len.cpu().numpy().astype(np.int64)
        offs = enc.d_pack_off.cpu().numpy().astype(np.int64)
        self.blocks = (_lib.Block * n)()
        for i in range(n):
            b = w["blocks"][i]
            self.blocks[i] = _lib.Block(int(offs[i]), b.in_off, int(lens[i]), b.in_len, 0, 0)
        self.d_raw = torch.zeros(w["d_in"].numel(), dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(n, dtype=torch.int32, device=dev)
        reads = w["reads"]
        self.mats_h = np.ascontiguousarray(enc.mats.cpu().numpy())
        self.st_b = torch.zeros(reads.n, dtype=torch.int32, device=dev)
        self.st_q = torch.zeros(reads.n, dtype=torch.int32, device=dev)
        self.reader = None
        if enc.writer is not None:
            self.reader = rec.CramRecordReader(eng, dev, reads.n, w["n_cont"], int(reads.bases.numel()), w["feat_cap"],
                                               16 * w["feat_cap"])
            n_r = reads.n
            z = lambda k, dt: torch.zeros(k, dtype=dt, device=dev)
            self.cig_cap = n_r + 2 * w["feat_cap"] + 1
            self.n_cigar, self.cigar_off, self.cigar = z(n_r, torch.int32), z(n_r + 1, torch.int64), z(self.cig_cap, torch.int32)
            self.aln_end, self.st_c, self.st_m = z(n_r, torch.int32), z(1, torch.int32), z(w["n_cont"], torch.int32)
            self.cont_ref_id = z(w["n_cont"], torch.int32)
            self.bam_cap = n_r * (36 + 12) + 4 * self.cig_cap + 2 * int(reads.bases.numel()) + 64
            self.bam, self.rec_off, self.st_bam = z(self.bam_cap + 16, torch.uint8), z(n_r + 1, torch.int64), z(n_r, torch.int32)
        else:  # the round-1 form: columns and features of the encode side
            self.bases = torch.zeros(reads.bases.numel() + 16, dtype=torch.uint8, device=dev)
            self.rflags = reads.rflags | 16  # CRAMGPU_RF_FORCE_QS: lossless records carry CF_FORCE_PRESERVE_QS
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        self.k2_ms, self.ms, self.bam_ms, self.gz_ms = [], [], [], []
        # G's stream has the lowest priority the device offers: K2 and G become ready at the same moment, and when G's
        # 10 000 workgroups reach the CUs first they keep the decoder's resident waves out for the length of G's streams
        # (K2 50 ms instead of 26: seen in two of eight launches of a profiled run)
        self.side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("BENCH_G_PRIORITY", "10"))) if str(dev).startswith("cuda") else None

    def run(self, timed=True):
        w, L, reads = self.w, _lib.lib(), self.w["reads"]
        s = torch.cuda.current_stream().cuda_stream
        if timed:
            self.ev[0].record()
        gz = self.reader is not None and getattr(self, "n_gz", 0)
        side = self.side if gz else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)  # what came before the leg; NOT the rANS decode launched next
        self.eng.decode_dev(self.enc.d_pack, self.d_raw, self.blocks, w["n_blocks"], self.d_len)
        if timed:
            self.ev[1].record()
        if gz:
            # G beside K2 on a stream of its own, launched after it: the decoder's resident grid takes its wave slots and its
            # share of the LDS first (it leaves a reserve: CRAMGPU_DEC_LDS_RESERVE), the gunzip workgroups fill the rest,
            # longest streams first.  A stream is a chain of table reads and branches: it needs few vector issue slots.
            if side is None:  # host build of the kernels (tests/test_glue_hostsim.py)
                self.eng.gunzip_dev(self.d_gz, self.d_gz_raw, self.d_gz_desc, self.n_gz, self.d_gz_len, stream=0)
            else:
                # ... and not before K2's kernel is the next thing in its stream (its block table is on the device): on a
                # backlogged GPU G's wait ends the moment K2's staging copies begin, and G would reach the CUs first
                _lib.check(L.cramgpu_stream_wait_staged(self.eng.handle, side.cuda_stream))
                with torch.cuda.stream(side):
                    self.eng.gunzip_dev(self.d_gz, self.d_gz_raw, self.d_gz_desc, self.n_gz, self.d_gz_len)
                    if timed:
                        self.ev[4].record()
                main.wait_stream(side)
        if self.reader is not None:
            rd = self.reader
            rd.read(self.arena_base, self.r_off, self.r_len, self.cont_first, self.series_bytes, multiref=self.multiref,
                    ap_base=self.ap_base, ap_delta=True)
            fs = rd.feats.struct()
            _lib.check(L.cramgpu_restore_bases_dev(
                self.eng.handle, reads.n, rd.read_len.data_ptr(), rd.aln_start.data_ptr(), rd.rflags.data_ptr(),
                reads.read_cont.data_ptr(), reads.n_cont, reads.win.data_ptr(), C.byref(fs), rd.seq.data_ptr(),
                rd.seq_off.data_ptr(), rd.seq_cap, w["ref"].data_ptr(), self.mats_h.ctypes.data, rd.bases.data_ptr(),
                rd.base_off.data_ptr(), self.st_b.data_ptr(), s))
            _lib.check(L.cramgpu_restore_quals_dev(self.eng.handle, reads.n, rd.read_len.data_ptr(), rd.rflags.data_ptr(),
                                                   C.byref(fs), 30, rd.quals.data_ptr(), rd.base_off.data_ptr(),
                                                   self.st_q.data_ptr(), s))
        else:
            fs = self.enc.feats.struct()
            _lib.check(L.cramgpu_restore_bases_dev(
                self.eng.handle, reads.n, reads.read_len.data_ptr(), reads.aln_start.data_ptr(), self.rflags.data_ptr(),
                reads.read_cont.data_ptr(), reads.n_cont, reads.win.data_ptr(), C.byref(fs), reads.bases.data_ptr(),
                reads.base_off.data_ptr(), int(reads.bases.numel()), w["ref"].data_ptr(), self.mats_h.ctypes.data,
                self.bases.data_ptr(),
                reads.base_off.data_ptr(), self.st_b.data_ptr(), s))
            # the qualities of a read sit in the decoded arena inside its container's QS block (q_off), restored in place
            _lib.check(L.cramgpu_restore_quals_dev(self.eng.handle, reads.n, reads.read_len.data_ptr(), self.rflags.data_ptr(),
                                                   C.byref(fs), 30, self.d_raw.data_ptr(), self.q_off.data_ptr(),
                                                   self.st_q.data_ptr(), s))
        if timed:
            self.ev[2].record()
            torch.cuda.synchronize()
            self.k2_ms.append(self.ev[0].elapsed_time(self.ev[1]))
            if gz and side is not None:
                self.gz_ms.append(self.ev[0].elapsed_time(self.ev[4]))
            self.ms.append(self.ev[0].elapsed_time(self.ev[2]))

    def assemble(self, timed=True):
        """A: what stands between the normalizer and BGZF in CramToBam_OBA_Function.java:64-75, on the reader's columns."""
        w, L, reads, rd = self.w, _lib.lib(), self.w["reads"], self.reader
        s = torch.cuda.current_stream().cuda_stream
        fs = rd.feats.struct()
        if timed:
            self.ev[3].record()
        cs = rd.cols.struct()
        _lib.check(L.cramgpu_cigars_dev(self.eng.handle, reads.n, rd.read_len.data_ptr(), rd.aln_start.data_ptr(),
                                        rd.cols.bam_flags.data_ptr(), C.byref(fs), self.n_cigar.data_ptr(),
                                        self.cigar_off.data_ptr(), self.cigar.data_ptr(), self.cig_cap, self.aln_end.data_ptr(),
                                        self.st_c.data_ptr(), s))
        _lib.check(L.cramgpu_restore_mates_dev(self.eng.handle, reads.n, w["n_cont"], self.cont_first.data_ptr(), C.byref(cs),
                                               self.cont_ref_id.data_ptr(), rd.aln_start.data_ptr(), self.aln_end.data_ptr(),
                                               self.st_m.data_ptr(), s))
        b = _lib.BamCols()
        b.ref_id, b.aln_start, b.aln_end, b.mapq = (rd.cols.ref_id.data_ptr(), rd.aln_start.data_ptr(), self.aln_end.data_ptr(),
                                                    rd.mapq.data_ptr())
        b.bam_flags, b.mate_flags, b.mate_ref = rd.cols.bam_flags.data_ptr(), rd.cols.mate_flags.data_ptr(), rd.cols.mate_ref.data_ptr()
        b.mate_start, b.tlen, b.read_len = rd.cols.mate_start.data_ptr(), rd.cols.tlen.data_ptr(), rd.read_len.data_ptr()
        b.base_off, b.bases, b.quals, b.qual_status = (rd.base_off.data_ptr(), rd.bases.data_ptr(), rd.quals.data_ptr(),
                                                       self.st_q.data_ptr())
        b.n_cigar, b.cigar_off, b.cigar, b.name_base = self.n_cigar.data_ptr(), self.cigar_off.data_ptr(), self.cigar.data_ptr(), 1
        _lib.check(L.cramgpu_bam_encode_dev(self.eng.handle, reads.n, C.byref(b), self.bam.data_ptr(), self.bam_cap,
                                            self.rec_off.data_ptr(), self.st_bam.data_ptr(), s))
        if timed:
            self.ev[2].record()
            torch.cuda.synchronize()
            self.bam_ms.append(self.ev[3].elapsed_time(self.ev[2]))

    # ---- the whole way out of a container, CramToBam_OBA_Function.java:53-83, in ONE timed region ----
    def run_full(self, timed):
        """K2 beside G -> R -> N1 / N2 -> A (CIGARs, mates, BAM records) -> the stripe as BGZF members (Z)."""
        reads = self.w["reads"]
        if not hasattr(self, "bgzf_img"):  # first (warm-up) call
            piece, slot = 65280, 65536
            nmax = (self.bam_cap + piece - 1) // piece
            self.bgzf_desc_h = np.zeros(nmax, BLOCK_DTYPE)
            self.bgzf_desc_h["in_off"] = np.arange(nmax, dtype=np.int64) * piece
            self.bgzf_desc_h["out_off"] = np.arange(nmax, dtype=np.int64) * slot
            self.bgzf_desc_h["out_cap"] = slot
            self.bgzf_desc = torch.zeros(nmax, BLOCK_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
            self.bgzf_img = torch.zeros(nmax * slot + 16, dtype=torch.uint8, device=self.dev)
            self.bgzf_len = torch.zeros(nmax, dtype=torch.int32, device=self.dev)
            self.ev_full = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            self.full_ms = []
        if timed:
            self.ev_full[0].record()
        self.run(False)
        self.assemble(False)
        size = int(self.rec_off[reads.n].item())  # the host needs the stripe's size anyway (it writes the file)
        piece = 65280
        n = (size + piece - 1) // piece
        d = self.bgzf_desc_h
        d["in_len"][:n] = np.minimum(piece, size - d["in_off"][:n])
        self.bgzf_desc[:n].copy_(torch.from_numpy(d[:n].view(np.uint8).reshape(n, -1)))
        self.eng.gzip_dev(self.bam, self.bgzf_img, self.bgzf_desc, n, self.bgzf_len, bgzf=True)
        self.bgzf_n, self.bam_bytes = n, size
        if timed:
            self.ev_full[1].record()
            torch.cuda.synchronize()
            self.full_ms.append(self.ev_full[0].elapsed_time(self.ev_full[1]))

    def verify_full(self):
        """Every BGZF member of run_full inflates (gunzip kernel) back to its piece of the stripe; four of them by zlib."""
        import zlib
        n, size, d = self.bgzf_n, self.bam_bytes, self.bgzf_desc_h
        lens = self.bgzf_len[:n].cpu().numpy()
        assert (lens > 0).all(), "a BGZF member failed"
        b = np.zeros(n, BLOCK_DTYPE)
        b["in_off"], b["out_off"], b["in_len"], b["out_cap"] = d["out_off"][:n], d["in_off"][:n], lens, d["in_len"][:n]
        d_back = torch.zeros(size + 16, dtype=torch.uint8, device=self.dev)
        d_blen = torch.zeros(n, dtype=torch.int32, device=self.dev)
        self.eng.gunzip_dev(self.bgzf_img, d_back, torch.as_tensor(b.view(np.uint8).copy()).to(self.dev), n, d_blen)
        torch.cuda.synchronize()
        assert np.array_equal(d_blen.cpu().numpy(), d["in_len"][:n]) and torch.equal(d_back[:size], self.bam[:size]), "BGZF image -> stripe"
        for k in (0, 1, n // 2, n - 1):
            m = self.bgzf_img[int(d["out_off"][k]):int(d["out_off"][k]) + int(lens[k])].cpu().numpy().tobytes()
            raw = self.bam[int(d["in_off"][k]):int(d["in_off"][k]) + int(d["in_len"][k])].cpu().numpy().tobytes()
            assert m[12:14] == b"BC" and zlib.decompress(m, 31) == raw, "BGZF member %d" % k
        return dict(bgzf_members=int(n), bam_bytes=int(size), bgzf_bytes=int(lens.sum()))

    # ---- the way IN, Bam_OBA_Supplier.java:45-82 + BAMRecordCodec#decode: BGZF members -> records in columns ----
    def run_ingest(self, timed):
        """The BGZF image run_full wrote, back into record columns in one region: G (every member inflated) -> record
        boundaries on the device -> BAM record decode.  What a BAM -> CRAM conversion does in front of K3."""
        from cramtools_amd import bam as bam_mod
        reads, n, size, d = self.w["reads"], self.bgzf_n, self.bam_bytes, self.bgzf_desc_h
        tot, ncg = int(reads.bases.numel()), int(reads.cigar.numel())
        if not hasattr(self, "ing_back"):
            lens = self.bgzf_len[:n].cpu().numpy()
            b = np.zeros(n, BLOCK_DTYPE)
            b["in_off"], b["out_off"], b["in_len"], b["out_cap"] = d["out_off"][:n], d["in_off"][:n], lens, d["in_len"][:n]
            self.ing_desc = torch.as_tensor(b.view(np.uint8).copy()).to(self.dev)
            self.ing_back = torch.zeros(size + 16, dtype=torch.uint8, device=self.dev)
            self.ing_len = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self.ing_codec = bam_mod.BAMRecordCodec(self.eng, self.dev)
            self.ing_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            self.ingest_ms = []
        if timed:
            self.ing_ev[0].record()
        self.eng.gunzip_dev(self.bgzf_img, self.ing_back, self.ing_desc, n, self.ing_len)
        off, n_rec, st = self.ing_codec.index_dev(self.ing_back, size, 1, reads.n + 1)
        assert st == 0 and n_rec == reads.n, "record boundaries of the inflated stripe"
        self.ing_cols = self.ing_codec.decode(self.ing_back, off, tot, ncg)
        if timed:
            self.ing_ev[1].record()
            torch.cuda.synchronize()
            self.ingest_ms.append(self.ing_ev[0].elapsed_time(self.ing_ev[1]))

    def verify_ingest(self):
        reads, dd = self.w["reads"], self.ing_cols
        tot, ncg = int(self.w["rs"].read_len.sum().item()), int(reads.n_cigar.sum().item())
        assert int((dd["status"] != 0).sum().item()) == 0, "the inflated stripe does not parse"
        assert torch.equal(dd["bases"][:tot], reads.bases[:tot]) and torch.equal(dd["quals"][:tot], reads.quals[:tot]), "ingest: seq / qual"
        assert torch.equal(dd["cigar"][:ncg], reads.cigar[:ncg]) and torch.equal(dd["aln_start"], reads.aln_start), "ingest: cigar / pos"
        return "every record of the BGZF image back in columns: bases, scores, CIGARs and starts equal the reads"

    def prepare(self):
        """Reader tables: where every series of every container lies (rANS series in the decoded arena, the others raw in
        the writer's arena), relative to one base address; or (round-1 form) per-read offsets of the quality bytes."""
        w, reads = self.w, self.w["reads"]
        ns = len(SERIES)
        if self.reader is not None:
            wr = self.enc.writer
            off, ln = wr.lengths()
            base = wr.arena.data_ptr()
            r_off = off.copy()
            in_off = np.array([w["blocks"][i].in_off for i in range(w["n_blocks"])], np.int64).reshape(w["n_cont"], ns)
            r_off[:, self.enc.ids] = self.d_raw.data_ptr() + in_off - base
            # the other series are GZIP blocks in a CRAM file (ExternalCompression#gzip, level 5: the host's job on encode):
            # deflated here once at set-up, inflated on the device inside the timed leg (G), read by R from where G puts them
            import zlib
            rest = [k for k in range(off.shape[1]) if k not in set(self.enc.ids.tolist())]
            arena_h = wr.arena.cpu().numpy()
            gz, desc, in_at, out_at = [], [], 0, 0
            self.gz_slots = []
            for c in range(w["n_cont"]):
                for k in rest:
                    n_raw = int(ln[c, k])
                    if n_raw == 0:
                        continue
                    co = zlib.compressobj(5, zlib.DEFLATED, 31)
                    z = co.compress(arena_h[int(off[c, k]):int(off[c, k]) + n_raw].tobytes()) + co.flush()
                    gz.append(z)
                    desc.append((in_at, out_at, len(z), n_raw))
                    self.gz_slots.append((c, k))
                    in_at += len(z)
                    out_at += (n_raw + 31) & ~15  # 16 readable bytes behind every stream, as the reader asks
            self.n_gz = len(gz)
            order = sorted(range(len(gz)), key=lambda i: -len(gz[i]))  # longest first: the launch is as long as its longest chain
            gz_sorted, desc_sorted, slots_sorted, in_at = [], [], [], 0
            for i in order:
                gz_sorted.append(gz[i])
                desc_sorted.append((in_at, desc[i][1], desc[i][2], desc[i][3]))
                slots_sorted.append(self.gz_slots[i])
                in_at += len(gz[i])
            gz, desc, self.gz_slots = gz_sorted, desc_sorted, slots_sorted
            if self.n_gz:
                blocks = (_lib.Block * self.n_gz)()
                for i, (a_, b_, c_, d_) in enumerate(desc):
                    blocks[i] = _lib.Block(a_, b_, c_, d_, 0, 0)
                    r_off[self.gz_slots[i][0], self.gz_slots[i][1]] = 0  # set below, once d_gz_raw exists
                self.d_gz = torch.as_tensor(np.frombuffer(b"".join(gz) + bytes(16), np.uint8).copy()).to(self.dev)
                self.d_gz_raw = torch.zeros(out_at + 16, dtype=torch.uint8, device=self.dev)
                self.d_gz_desc = torch.as_tensor(np.frombuffer(bytes(blocks), np.uint8).copy()).to(self.dev)
                self.d_gz_len = torch.zeros(self.n_gz, dtype=torch.int32, device=self.dev)
                self.gz_raw_len = np.array([d[3] for d in desc], np.int32)
                self.gz_bytes = (in_at, int(self.gz_raw_len.sum()))
                for i, (a_, b_, c_, d_) in enumerate(desc):
                    r_off[self.gz_slots[i][0], self.gz_slots[i][1]] = self.d_gz_raw.data_ptr() + b_ - base
            self.arena_base = base
            self.r_off = torch.as_tensor(r_off.reshape(-1)).to(self.dev)
            self.r_len = torch.as_tensor(ln.reshape(-1).copy()).to(self.dev)
            self.series_bytes = int(ln.sum()) + 64
            self.cont_first = torch.as_tensor(np.minimum(np.arange(w["n_cont"] + 1, dtype=np.int64) * READS_PER_CONTAINER,
                                                         reads.n).astype(np.int32)).to(self.dev)
            self.multiref = w["cols"].multiref
            self.ap_base = reads.aln_start[self.cont_first[:-1].to(torch.int64)].contiguous()
            return
        qs_off = torch.as_tensor(np.array([w["blocks"][c * ns].in_off for c in range(w["n_cont"])], np.int64), device=self.dev)
        first = reads.base_off[torch.arange(w["n_cont"], device=self.dev) * READS_PER_CONTAINER]
        cont = reads.read_cont.to(torch.int64)
        self.q_off = qs_off[cont] + (reads.base_off - first[cont])

    def gzip_series(self, reps=2, piece=0):
        """The encode half of the GZIP series (ExternalCompression#gzip; outside every timed region): the raw streams the
        leg inflated are written as gzip members by the device (cramgpu_gzip_dev), inflated again by the gunzip kernel and
        compared; sizes beside the zlib level 5 members made at set-up.  piece > 0: series longer than that are cut into
        pieces of that many bytes, one member each (GZIPInputStream reads members written back to back as one stream):
        a call lasts as long as its longest stream."""
        if not getattr(self, "n_gz", 0):
            return None
        old = np.frombuffer(self.d_gz_desc.cpu().numpy().tobytes(), BLOCK_DTYPE)
        step = piece if piece > 0 else 1 << 30
        src, ln = [], []
        for o, n_raw in zip(old["out_off"].tolist(), old["out_cap"].tolist()):
            for at in range(0, n_raw, step):
                src.append(o + at)
                ln.append(min(step, n_raw - at))
        n = len(src)
        enc = np.zeros(n, BLOCK_DTYPE)
        enc["in_off"], enc["in_len"] = src, ln
        enc["out_cap"] = [xc.Engine.gzip_bound(x) for x in ln]
        enc["out_off"] = np.concatenate([[0], np.cumsum(enc["out_cap"].astype(np.int64))[:-1]])
        at = int(enc["out_cap"].astype(np.int64).sum())
        order = np.argsort(-enc["in_len"], kind="stable")  # longest first
        enc = enc[order]
        d_enc = torch.as_tensor(enc.view(np.uint8).copy()).to(self.dev)
        d_z = torch.zeros(at + 16, dtype=torch.uint8, device=self.dev)
        d_zlen = torch.zeros(n, dtype=torch.int32, device=self.dev)
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.eng.gzip_dev(self.d_gz_raw, d_z, d_enc, n, d_zlen)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        zl = d_zlen.cpu().numpy()
        assert (zl > 0).all(), "a GZIP series failed to compress"
        back = np.zeros(n, BLOCK_DTYPE)
        back["in_off"], back["out_off"], back["in_len"], back["out_cap"] = enc["out_off"], enc["in_off"], zl, enc["in_len"]
        d_back = torch.zeros_like(self.d_gz_raw)
        d_blen = torch.zeros(n, dtype=torch.int32, device=self.dev)
        self.eng.gunzip_dev(d_z, d_back, torch.as_tensor(back.view(np.uint8).copy()).to(self.dev), n, d_blen)
        torch.cuda.synchronize()
        assert np.array_equal(d_blen.cpu().numpy(), enc["in_len"]) and torch.equal(d_back, self.d_gz_raw), "gzip -> gunzip round trip"
        return dict(ms=float(min(ms)), streams=self.n_gz, members=n, raw_bytes=self.gz_bytes[1], gz_bytes=int(zl.sum()),
                    zlib_level5_bytes=self.gz_bytes[0])

    def verify(self):
        w, reads = self.w, self.w["reads"]
        n = w["d_in"].numel() - 16
        assert bool((self.d_len.cpu().numpy() == np.array([w["blocks"][i].in_len for i in range(w["n_blocks"])])).all()), \
            "a block failed to decode"
        assert torch.equal(self.d_raw[:n], w["d_in"][:n]), "decoded blocks differ from the raw arena"
        tot = int(w["rs"].read_len.sum().item())
        assert int((self.st_b != 0).sum().item()) == 0, "restoreReadBases reported a status"
        assert int((self.st_q != 0).sum().item()) == 0, "restoreQualityScores reported a status"
        if self.reader is not None:
            if getattr(self, "n_gz", 0):
                assert np.array_equal(self.d_gz_len.cpu().numpy(), self.gz_raw_len), "a GZIP block failed to inflate"
            rd = self.reader.check()
            assert torch.equal(rd.read_len, reads.read_len) and torch.equal(rd.aln_start, reads.aln_start), "RL / AP"
            assert torch.equal(rd.base_off[:-1], reads.base_off), "record offsets"
            for k in rec.RECORD_COLS:
                assert torch.equal(getattr(rd.cols, k), getattr(w["cols"], k)), k
            nf = self.enc.feats.count
            assert rd.feats.count == nf and torch.equal(rd.feats.off, self.enc.feats.off), "feature counts"
            assert torch.equal(rd.feats.op[:nf], self.enc.feats.op[:nf]) and torch.equal(rd.feats.pos[:nf], self.enc.feats.pos[:nf])
            assert torch.equal(rd.bases[:tot], reads.bases[:tot]), "restored bases differ from the reads"
            assert torch.equal(rd.quals[:tot], reads.quals[:tot]), "restored scores differ from the reads"
            return ("all %d blocks and %d reads round-trip through writer, rANS, reader and normalizer (decoded arena == raw "
                    "arena, record columns, features, bases and scores == the reads)" % (w["n_blocks"], reads.n))
        assert torch.equal(self.bases[:tot], reads.bases[:tot]), "restored bases differ from the reads"
        return "all %d blocks and %d reads round-trip (decoded arena == raw arena, restored bases == read bases)" % (
            w["n_blocks"], reads.n)


    def verify_assembly(self):
        """The BAM stripe: CIGARs rebuilt from the features are the reads' own, and the stripe decodes back to the reads."""
        w, reads = self.w, self.w["reads"]
        tot = int(w["rs"].read_len.sum().item())
        from cramtools_amd import bam as bam_mod
        ncg = int(reads.n_cigar.sum().item())
        assert int(self.st_c.item()) == 0 and int((self.st_m != 0).sum().item()) == 0, "CIGAR / mate status"
        assert int((self.st_bam != 0).sum().item()) == 0, "a BAM record reported a status"
        assert torch.equal(self.n_cigar, reads.n_cigar) and torch.equal(self.cigar[:ncg], reads.cigar[:ncg]), \
            "CIGARs rebuilt from the features differ from the reads' CIGARs"
        size = int(self.rec_off[reads.n].item())
        codec = bam_mod.BAMRecordCodec(self.eng, self.dev)
        d = codec.decode(self.bam, self.rec_off, tot, ncg)  # warm-up (allocates its outputs)
        torch.cuda.synchronize()
        del d
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d = codec.decode(self.bam, self.rec_off, tot, ncg)
        e1.record()
        torch.cuda.synchronize()
        self.bam_decode_ms = e0.elapsed_time(e1)  # includes the zero-fill of the output tensors the wrapper allocates
        assert int((d["status"] != 0).sum().item()) == 0, "the BAM stripe does not parse"
        assert torch.equal(d["bases"][:tot], reads.bases[:tot]) and torch.equal(d["quals"][:tot], reads.quals[:tot]), "BAM seq / qual"
        assert torch.equal(d["cigar"][:ncg], reads.cigar[:ncg]) and torch.equal(d["aln_start"], reads.aln_start), "BAM cigar / pos"
        assert torch.equal(d["bam_flags"] & ~0x28, w["cols"].bam_flags & 0xFFD7), "BAM flags"
        self.bam_bytes = size
        # record boundaries of the stripe on the device (cramgpu_bam_index_dev) against the offsets the encoder laid down
        codec.index_dev(self.bam, size, 1, reads.n + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        off, n_rec, st = codec.index_dev(self.bam, size, 1, reads.n + 1)
        self.bam_index_ms = (time.perf_counter() - t0) * 1e3  # host clock around the call and its two scalar reads
        assert st == 0 and n_rec == reads.n and torch.equal(off, self.rec_off), "record boundaries found on the device"
        return ("; the %d-byte BAM stripe built from them (CIGARs from features, mate restoration, BAM records) decodes back to "
                "the reads" % size)


def bgzf_image(dec):
    """SURVEY 8(f) row 3, the last step of the way out: the BAM stripe cut into 65280-byte pieces (htsjdk's
    BlockCompressedOutputStream block size) and written as BGZF members on the device (cramgpu_gzip_dev); every member is
    inflated again by the gunzip kernel and compared with the stripe, a sample by zlib on the host.  Outside every timed
    region of the headline."""
    import zlib
    assert C.sizeof(_lib.Block) == BLOCK_DTYPE.itemsize
    size, piece, slot = dec.bam_bytes, 65280, 65536
    n = (size + piece - 1) // piece
    d = np.zeros(n, BLOCK_DTYPE)
    d["in_off"] = np.arange(n, dtype=np.int64) * piece
    d["out_off"] = np.arange(n, dtype=np.int64) * slot
    d["in_len"] = np.minimum(piece, size - d["in_off"])
    d["out_cap"] = slot
    d_desc = torch.as_tensor(d.view(np.uint8)).to(dec.dev)
    d_img = torch.zeros(n * slot + 16, dtype=torch.uint8, device=dec.dev)
    d_len = torch.zeros(n, dtype=torch.int32, device=dec.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec.eng.gzip_dev(dec.bam, d_img, d_desc, n, d_len, bgzf=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lens = d_len.cpu().numpy()
    assert (lens > 0).all(), "a BGZF member failed: %r" % (lens[lens <= 0][:4],)
    b = np.zeros(n, BLOCK_DTYPE)
    b["in_off"], b["out_off"], b["in_len"], b["out_cap"] = d["out_off"], d["in_off"], lens, d["in_len"]
    d_back = torch.zeros(size + 16, dtype=torch.uint8, device=dec.dev)
    d_blen = torch.zeros(n, dtype=torch.int32, device=dec.dev)
    dec.eng.gunzip_dev(d_img, d_back, torch.as_tensor(b.view(np.uint8)).to(dec.dev), n, d_blen)
    torch.cuda.synchronize()
    assert np.array_equal(d_blen.cpu().numpy(), d["in_len"]) and torch.equal(d_back[:size], dec.bam[:size]), "BGZF image -> stripe"
    for k in (0, 1, n // 2, n - 1):
        m = d_img[int(d["out_off"][k]):int(d["out_off"][k]) + int(lens[k])].cpu().numpy().tobytes()
        raw = dec.bam[int(d["in_off"][k]):int(d["in_off"][k]) + int(d["in_len"][k])].cpu().numpy().tobytes()
        assert m[12:14] == b"BC" and int.from_bytes(m[16:18], "little") == len(m) - 1 and zlib.decompress(m, 31) == raw, "BGZF member %d" % k
    return dict(ms=ms, blocks=int(n), bam_bytes=int(size), bgzf_bytes=int(lens.sum()), GBps_in=size / ms / 1e6,
                note="the stripe as BGZF members of 65280 input bytes (cramgpu_gzip_dev, bgzf); all members inflated again by the "
                     "gunzip kernel and compared with the stripe, four of them by zlib")


def host_threads():
    """Threads of the CPU legs: every core this process may use.  The GPU boxes show the whole machine in os.cpu_count()
    but give a job a share of it, exported as OMP_NUM_THREADS (16 per GPU); elsewhere it is os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return n


def cpu_baseline(w, budget_s=12.0):
    """The CPU oracle (C restatement of htsjdk 2.4.1 rANS, oracle/rans4x8.c) on a bounded sample of the same blocks:
    encode (orc_rans_compress_many) and decode (orc_rans_uncompress_many) of the same containers, one call per thread
    (one workspace of tables per thread), containers dealt round-robin to the threads."""
    import shutil
    from oracle import orc
    orc.build()
    L = orc.lib()
    cores = host_threads()
    n_series = len(SERIES)
    blocks, d_in = w["blocks"], w["d_in"]

    def take(c0, c1):
        lo = blocks[c0 * n_series].in_off
        last = blocks[c1 * n_series - 1]
        hi = last.in_off + last.in_len
        arena = d_in[lo:hi].cpu().numpy()
        off = np.array([blocks[i].in_off - lo for i in range(c0 * n_series, c1 * n_series)], np.int64)
        ln = np.array([blocks[i].in_len for i in range(c0 * n_series, c1 * n_series)], np.int32)
        od = np.array([blocks[i].order for i in range(c0 * n_series, c1 * n_series)], np.int32)
        return arena, off, ln, od

    def p(a):
        return a.ctypes.data_as(C.c_void_p)

    def threaded(fn, shares):
        ths = [threading.Thread(target=fn, args=(sh,)) for sh in shares if sh.size]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0

    def run(arena, off, ln, od, threads):
        caps = np.array([orc.rans_bound(int(x)) for x in ln], np.int64)
        ooff = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.int64)
        out = np.empty(int(caps.sum()) + 16, np.uint8)
        olen = np.zeros(ln.size, np.int32)
        n_c = ln.size // n_series
        shares = [(np.arange(t, n_c, threads)[:, None] * n_series + np.arange(n_series)[None, :]).reshape(-1)
                  for t in range(threads)]  # block indices of thread t: containers t, t + threads, ...

        def enc(idx):
            ol = np.zeros(idx.size, np.int32)
            r = L.orc_rans_compress_many(p(arena), p(np.ascontiguousarray(off[idx])), p(np.ascontiguousarray(ln[idx])),
                                         p(np.ascontiguousarray(od[idx])), C.c_int32(idx.size), p(out),
                                         p(np.ascontiguousarray(ooff[idx])), p(ol))
            assert r >= 0
            olen[idx] = ol
        dt_enc = threaded(enc, shares)
        # decode leg: the streams just made, back into a second arena laid out like the raw one
        back = np.empty(arena.size + 16, np.uint8)
        blen = np.zeros(ln.size, np.int32)

        def dec(idx):
            bl = np.zeros(idx.size, np.int32)
            r = L.orc_rans_uncompress_many(p(out), p(np.ascontiguousarray(ooff[idx])), p(np.ascontiguousarray(olen[idx])),
                                           C.c_int32(idx.size), p(back), p(np.ascontiguousarray(off[idx])),
                                           p(np.ascontiguousarray(ln[idx])), p(bl))
            assert r >= 0
            blen[idx] = bl
        dt_dec = threaded(dec, shares)
        assert np.array_equal(blen, ln)
        for i in range(ln.size):
            assert np.array_equal(back[off[i]:off[i] + ln[i]], arena[off[i]:off[i] + ln[i]]), "oracle round trip, block %d" % i
        return dt_enc, dt_dec, out, ooff, olen

    a, o, l, d = take(0, 1)
    t1, t1d, _, _, _ = run(a, o, l, d, 1)  # pilot on one container to size the sample
    per_cont = max(t1 + t1d, 1e-4)
    n_c = int(min(w["n_cont"], max(cores, budget_s * cores / per_cont)))
    n_c = max(cores, n_c - n_c % cores) if w["n_cont"] >= cores else w["n_cont"]
    n_c = min(n_c, w["n_cont"])
    a, o, l, d = take(0, n_c)
    dt, dt_dec, out, ooff, olen = run(a, o, l, d, min(cores, n_c))
    mb = float(l.sum()) / 1e6
    return dict(value=mb / dt, unit="MB/s", cores=min(cores, n_c), nproc=os.cpu_count(), kind="port",
                jvm=shutil.which("java") is not None, decode_value=mb / dt_dec, encode_s=dt, decode_s=dt_dec,
                sample="oracle/rans4x8.c (C restatement of htsjdk 2.4.1 RANS; no JVM on this host when jvm is false) on the "
                       "first %d containers (%d blocks, %.1f MB raw): encode, then decode of the same streams; one call and "
                       "one table workspace per thread, containers dealt round-robin to %d threads (all the cores this "
                       "process is given)" % (n_c, l.size, mb, min(cores, n_c))), (out, ooff, olen, n_c)


def pcie_inclusive(eng, w, per_job=256, n_cont=None, outstanding=4, reps=2, packed=True):
    """PCIe-inclusive rate of the asynchronous host-arena boundary (cramgpu_rans_submit / cramgpu_job_wait, what a JNI
    host uses): the containers' raw blocks start in pinned HOST arenas and the streams end there; `per_job` containers
    per submission, `outstanding` jobs in flight; packed = cramgpu_rans_submit_packed (streams stay packed in the host
    arena, the form a JNI host wants) instead of cramgpu_rans_submit (streams spread to their slots by a host memmove).
    Never the headline `value` (inputs resident in HBM)."""
    L = _lib.lib()
    ns = len(SERIES)
    n_cont = min(n_cont or w["n_cont"], w["n_cont"])
    d_in = w["d_in"]
    jobs = []
    for j0 in range(0, n_cont, per_job):
        idx = range(j0 * ns, min(j0 + per_job, n_cont) * ns)
        lens = [w["blocks"][i].in_len for i in idx]
        orders = [w["blocks"][i].order for i in idx]
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode(lens, orders)
        h_in, h_out, h_len = L.cramgpu_host_alloc(in_bytes), L.cramgpu_host_alloc(out_bytes), L.cramgpu_host_alloc(4 * len(lens))
        h_pos = L.cramgpu_host_alloc(8 * (len(lens) + 1))
        assert h_in and h_out and h_len and h_pos, "cramgpu_host_alloc failed"
        arena = torch.from_numpy(np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(in_bytes,)))
        # the job's input arena has the same back-to-back layout as the slice of the device arena it comes from
        b0, bl = w["blocks"][idx[0]], w["blocks"][idx[-1]]
        assert blocks[len(lens) - 1].in_off == bl.in_off - b0.in_off
        arena[:bl.in_off - b0.in_off + bl.in_len] = d_in[b0.in_off:bl.in_off + bl.in_len].cpu()
        jobs.append((blocks, in_bytes, out_bytes, h_in, h_out, h_len, len(lens), sum(lens), h_pos))

    def run():
        pending, raw, comp = [], 0, 0
        t0 = time.perf_counter()
        for (blocks, in_bytes, out_bytes, h_in, h_out, h_len, n, rb, h_pos) in jobs:
            if len(pending) == outstanding:
                j, hl, nn = pending.pop(0)
                _lib.check(L.cramgpu_job_wait(j))
                comp += int(np.ctypeslib.as_array(C.cast(hl, C.POINTER(C.c_int32)), shape=(nn,)).sum())
            if packed:
                j = L.cramgpu_rans_submit_packed(eng.handle, h_in, in_bytes, h_out, out_bytes, C.addressof(blocks), n, h_len, h_pos)
            else:
                j = L.cramgpu_rans_submit(eng.handle, 0, h_in, in_bytes, h_out, out_bytes, C.addressof(blocks), n, h_len)
            assert j, _lib.last_error()
            pending.append((j, h_len, n))
            raw += rb
        for j, hl, nn in pending:
            _lib.check(L.cramgpu_job_wait(j))
            comp += int(np.ctypeslib.as_array(C.cast(hl, C.POINTER(C.c_int32)), shape=(nn,)).sum())
        return time.perf_counter() - t0, raw, comp
    run()
    best = min(run() for _ in range(reps))
    # spot check: the first job's first stream came back where the synchronous path puts it
    blocks, _, _, h_in, h_out, h_len, n, _, _ = jobs[0]
    first_len = int(np.ctypeslib.as_array(C.cast(h_len, C.POINTER(C.c_int32)), shape=(n,))[0])
    head = bytes(np.ctypeslib.as_array(C.cast(h_out, C.POINTER(C.c_uint8)), shape=(9,)))
    assert head[0] == blocks[0].order and int.from_bytes(head[5:9], "little") == blocks[0].in_len and first_len > 9
    for (_, _, _, h_in, h_out, h_len, _, _, h_pos) in jobs:
        L.cramgpu_host_free(h_in), L.cramgpu_host_free(h_out), L.cramgpu_host_free(h_len), L.cramgpu_host_free(h_pos)
    dt, raw, comp = best
    return dict(value=raw / dt / 1e6, unit="MB/s", entry_point="cramgpu_rans_submit_packed" if packed else "cramgpu_rans_submit",
                containers_per_job=per_job, jobs_outstanding=outstanding,
                containers=n_cont, ms=dt * 1e3, raw_MB=raw / 1e6, compressed_MB=comp / 1e6,
                note="raw blocks from pinned host arenas, streams back into pinned host arenas (H2D + kernels + D2H)")


def pcie_coalesced(eng, w, n_cont=None, threads=(), tickets=(64, 128), max_batches=0, reps=2):
    """PCIe-inclusive rate of the coalescing queue at the reference's own job size — ONE container per submission
    (paralell/TransformerJob.java:29-41): raw blocks in pinned host arenas -> packed streams in pinned host arenas.
    `threads`: T worker threads that call the blocking cramgpu_rans_coalesced in a loop (the Java thread model);
    `tickets`: one thread keeps W tickets outstanding (cramgpu_rans_enqueue / cramgpu_ticket_wait).  Every stream's
    header is checked against its block.  Returns one dict per case.  Never the headline `value`."""
    import threading
    L = _lib.lib()
    ns = len(SERIES)
    n_cont = min(n_cont or w["n_cont"], w["n_cont"])
    d_in = w["d_in"]
    conts = []
    for k in range(n_cont):
        idx = range(k * ns, (k + 1) * ns)
        lens = [w["blocks"][i].in_len for i in idx]
        orders = [w["blocks"][i].order for i in idx]
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode(lens, orders)
        h_in, h_out = L.cramgpu_host_alloc(in_bytes), L.cramgpu_host_alloc(out_bytes)
        h_len, h_pos = L.cramgpu_host_alloc(4 * ns), L.cramgpu_host_alloc(8 * (ns + 1))
        assert h_in and h_out and h_len and h_pos, "cramgpu_host_alloc failed"
        arena = torch.from_numpy(np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(in_bytes,)))
        b0, bl = w["blocks"][idx[0]], w["blocks"][idx[-1]]
        arena[:bl.in_off - b0.in_off + bl.in_len] = d_in[b0.in_off:bl.in_off + bl.in_len].cpu()
        conts.append((blocks, in_bytes, out_bytes, h_in, h_out, h_len, h_pos, sum(lens)))
    raw = sum(c[-1] for c in conts)
    q = L.cramgpu_coalescer_open(eng.handle, max_batches)
    assert q, _lib.last_error()

    def call_args(c):
        blocks, in_bytes, out_bytes, h_in, h_out, h_len, h_pos, _ = c
        return (q, 0, h_in, in_bytes, h_out, out_bytes, C.addressof(blocks), ns, h_len, h_pos)

    def check(c, k):
        blocks, _, _, _, h_out, h_len, h_pos, _ = c
        lens = np.ctypeslib.as_array(C.cast(h_len, C.POINTER(C.c_int32)), shape=(ns,))
        pos = np.ctypeslib.as_array(C.cast(h_pos, C.POINTER(C.c_int64)), shape=(ns + 1,))
        out = np.ctypeslib.as_array(C.cast(h_out, C.POINTER(C.c_uint8)), shape=(int(pos[ns]),))
        assert (lens > 9).all(), (k, lens)
        for i in range(ns):
            head = bytes(out[int(pos[i]):int(pos[i]) + 9])
            assert head[0] == blocks[i].order and int.from_bytes(head[5:9], "little") == blocks[i].in_len, (k, i)
            assert int.from_bytes(head[1:5], "little") + 9 == lens[i], (k, i)
        return int(lens.sum())

    # worker threads: native ones (scripts/ubench/libco_driver.so, built by __graft_entry__.build) so that the figure
    # is the library's and not the interpreter lock's; Python threads when the helper is missing
    drv_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "ubench", "libco_driver.so")
    drv = C.CDLL(drv_path) if os.path.exists(drv_path) else None
    if drv is not None:
        drv.co_driver_threads.restype = C.c_double
        PP = C.c_void_p * n_cont
        I64A = C.c_int64 * n_cont
        arr = dict(h_in=PP(*[c[3] for c in conts]), in_bytes=I64A(*[c[1] for c in conts]), h_out=PP(*[c[4] for c in conts]),
                   out_bytes=I64A(*[c[2] for c in conts]), blocks=PP(*[C.addressof(c[0]) for c in conts]),
                   h_len=PP(*[c[5] for c in conts]), h_pos=PP(*[c[6] for c in conts]))

    def threads_run(T):
        if drv is not None:
            err = C.c_int32(0)
            dt = drv.co_driver_threads(C.cast(L.cramgpu_rans_coalesced, C.c_void_p), C.c_void_p(q), C.c_int32(n_cont), C.c_int32(T),
                                       arr["h_in"], arr["in_bytes"], arr["h_out"], arr["out_bytes"], arr["blocks"], C.c_int32(ns),
                                       arr["h_len"], arr["h_pos"], C.byref(err))
            assert err.value == 0, (err.value, _lib.last_error())
            return dt
        nxt, lock, errs = [0], threading.Lock(), []

        def worker():
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n_cont:
                    return
                rc = L.cramgpu_rans_coalesced(*call_args(conts[k]))
                if rc:
                    errs.append((k, rc, _lib.last_error()))
                    return

        ths = [threading.Thread(target=worker) for _ in range(T)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        assert not errs, errs[:3]
        return dt

    def window_run(W):
        pend = []
        t0 = time.perf_counter()
        for c in conts:
            if len(pend) == W:
                _lib.check(L.cramgpu_ticket_wait(q, pend.pop(0)))
            t = L.cramgpu_rans_enqueue(*call_args(c))
            assert t > 0, _lib.last_error()
            pend.append(t)
        for t in pend:
            _lib.check(L.cramgpu_ticket_wait(q, t))
        return time.perf_counter() - t0

    def stats():
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(L.cramgpu_coalesce_stats(q, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value

    window_run(64)  # warm-up: the batch slots grow to their working size
    res = []
    for mode, counts in (("threads", threads), ("tickets", tickets)):
        for n in counts:
            s0 = stats()
            dt = min((threads_run if mode == "threads" else window_run)(n) for _ in range(reps))
            s1 = stats()
            comp = sum(check(c, k) for k, c in enumerate(conts))
            res.append(dict(value=raw / dt / 1e6, unit="MB/s",
                            entry_point="cramgpu_rans_coalesced" if mode == "threads" else "cramgpu_rans_enqueue",
                            containers_per_submission=1, outstanding=n,
                            driver=("native threads" if drv is not None else "python threads") if mode == "threads" else mode,
                            containers=n_cont, ms=dt * 1e3,
                            raw_MB=raw / 1e6, compressed_MB=comp / 1e6,
                            containers_per_launch=(s1[1] - s0[1]) / max(1, s1[0] - s0[0])))
    L.cramgpu_coalescer_close(q)
    for c in conts:
        for p in c[3:7]:
            L.cramgpu_host_free(p)
    return res


# ======================================================================================================
# --config 4: BASELINE.json configs[4] (10 kbp reads, high indel rate, lossy model R8-N40-D8, order-1 rANS) as one timed
# pipeline.  Not the headline (that stays configs[1]); here so that the config's number is reproducible.
# ======================================================================================================
def long_read_set(rng, n, L, ref):
    """n start-sorted reads of exactly L bases: match runs (geometric, mean 12.5) with 3 %
    substitutions, cut by an insertion or a deletion of 1-3 bases each, uniform scores 1-30.  Vectorised numpy: the reads
    of one container are 10^8 bases.  Returns flat arrays shaped like synth.ReadSet."""
    K = int(L / 13.0 * 1.25) + 64
    run = rng.geometric(0.08, (n, K)).astype(np.int64)
    ins = rng.random((n, K)) < 0.5
    k = rng.integers(1, 4, (n, K)).astype(np.int64)
    used = run + np.where(ins, k, 0)  # read bases of a segment: its match run and the insertion behind it
    cum = np.cumsum(used, 1)
    assert (cum[:, -1] >= L).all()
    last = (cum < L).sum(1)  # the segment that reaches L: its run is cut to fit and nothing follows it
    j = np.arange(K)[None, :]
    before = np.where(last > 0, cum[np.arange(n), np.maximum(last - 1, 0)], 0)
    run[np.arange(n), last] = L - before
    valid = j <= last[:, None]
    indel = valid & (j < last[:, None])
    refspan = np.where(valid, run, 0) + np.where(indel & ~ins, k, 0)
    span = refspan.sum(1)
    starts = np.sort(rng.integers(1, ref.size - int(span.max()) - 16, n)).astype(np.int64)
    ref_at = starts[:, None] - 1 + np.cumsum(refspan, 1) - refspan  # 0-based reference position of each match run
    # CIGAR: M run, then I k / D k behind every segment but the last
    ops = np.stack([(run << 4) | 0, (k << 4) | np.where(ins, 1, 2)], 2)
    keep = np.stack([valid, indel], 2)
    cigar = ops[keep].astype(np.uint32)
    n_cigar = keep.reshape(n, -1).sum(1).astype(np.int32)
    # pieces of the read in order: match run (copied from the reference), insertion (random bases)
    plen = np.stack([np.where(valid, run, 0), np.where(indel & ins, k, 0)], 2).reshape(-1)
    psrc = np.stack([ref_at, np.full((n, K), -1, np.int64)], 2).reshape(-1)
    nz = plen > 0
    plen, psrc = plen[nz], psrc[nz]
    total = n * L
    assert int(plen.sum()) == total
    pstart = np.cumsum(plen) - plen
    within = np.arange(total, dtype=np.int64) - np.repeat(pstart, plen)
    src = np.repeat(psrc, plen)
    is_match = src >= 0
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, total)]
    bases[is_match] = ref[(src + within)[is_match]]
    sub = is_match & (rng.random(total) < 0.03)
    bases[sub] = acgt[(np.searchsorted(acgt, bases[sub]) + rng.integers(1, 4, int(sub.sum()))) % 4]
    quals = rng.integers(1, 31, total, dtype=np.uint8)
    return dict(n=n, bases=bases, quals=quals, base_off=np.arange(n, dtype=np.int64) * L, read_len=np.full(n, L, np.int32),
                aln_start=starts.astype(np.int32), aln_span=(span - 1).astype(np.int32), cigar=cigar,  # alignmentEnd - alignmentStart
                cigar_off=(np.cumsum(n_cigar) - n_cigar).astype(np.int64), n_cigar=n_cigar,
                mapq=np.where(rng.random(n) < 0.05, 0, 60).astype(np.uint8))


class StepC5:
    """One pass of configs[4]'s encode path over the resident reads: K3 -> F3 -> K5 (lossy model) -> merge of the Q
    features -> data-series writer (QS from the model's scores) -> rANS over the ten series -> gather."""

    def __init__(self, dev, eng, ref, rs, spec, n_cont):
        self.dev, self.eng, self.spec = dev, eng, spec
        n = rs["n"]
        pad = np.zeros(16, np.uint8)
        windows = [(0, ref.numel() - 16, 0)] * n_cont
        read_cont = (np.arange(n) // READS_PER_CONTAINER).astype(np.int32)
        mk = lambda q: rec.ReadBatch(dev, np.concatenate([rs["bases"], pad]), rs["base_off"], rs["read_len"], rs["aln_start"],
                                     np.concatenate([rs["cigar"], np.zeros(1, np.uint32)]).view(np.int32), rs["cigar_off"],
                                     rs["n_cigar"], quals=q, mapq=rs["mapq"], read_cont=read_cont, windows=windows)
        self.reads = mk(np.concatenate([rs["quals"], pad]))
        self.lossy_reads = mk(None)  # the same reads with the model's scores in place of the input scores
        self.ref, self.n, self.n_cont = ref, n, n_cont
        self.fac = rec.Sam2CramRecordFactory(eng)
        self.fac.set_ref_bases(ref)
        self.qsp = rec.QualityScorePreservation(eng, spec)
        self.wpos = self.qsp.tracks_windows(rs["aln_start"], rs["aln_span"], np.zeros(n, np.uint8), rs["mapq"], ref.numel() - 16)
        self.tot = int(rs["read_len"].sum())
        self.feat_cap = self.tot // 4
        first = np.arange(n_cont) * READS_PER_CONTAINER
        self.cols_np = dict(bam_flags=np.zeros(n, np.int32), read_group=np.full(n, -1, np.int32), mate_ref=np.full(n, -1, np.int32))
        self.ap_base = rs["aln_start"][first].astype(np.int32)
        self.ids = np.array([_lib.SERIES_ID[name] for name, _ in SERIES])
        self.writer = rec.Writer(eng, dev, n_cont, 3 * self.tot + 24 * (self.feat_cap + self.tot // 8) + 70 * n + 4096 * n_cont)
        nb = n_cont * len(SERIES)
        self.nb = nb
        self.blk = np.zeros(nb, BLOCK_DTYPE)
        self.blk["order"] = np.tile(np.array([o for _, o in SERIES], np.int32), n_cont)
        cap_total = int(1.06 * (self.tot + 40 * n + 8 * self.feat_cap)) + nb * (257 * 257 * 3 + 64)
        self.d_out = torch.empty(cap_total, dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(nb, dtype=torch.int32, device=dev)
        self.d_pos = torch.zeros(nb, dtype=torch.int64, device=dev)
        self.d_pack = torch.empty(self.tot + (64 << 20), dtype=torch.uint8, device=dev)
        self.d_pack_off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.enc_ms, self.raw_bytes = [], 0

    def run(self, timed):
        reads = self.reads
        feats, hist = self.fac.create_variations(reads, feature_cap=self.feat_cap)
        mats = np.stack([rec.substitution_matrix(hist[c]) for c in range(self.n_cont)])
        self.fac.assign_codes(reads, feats, mats)
        scores, force, status = self.qsp.add_quality_scores(reads, self.ref, self.wpos)
        merged = self.qsp.merge_quality_features(reads, scores, force, status, feats)
        self.lossy_reads.quals = scores
        cols = rec.RecordColumns(self.dev, self.n, ap_base=self.ap_base, comp_flags=2 | force.to(torch.int32), **self.cols_np)
        self.writer.write(self.lossy_reads, cols, merged)
        off, ln = self.writer.lengths()
        blk = self.blk
        blk["in_off"], blk["in_len"] = off[:, self.ids].reshape(-1), ln[:, self.ids].reshape(-1)
        cap = (1.05 * blk["in_len"].astype(np.float64) + 257.0 * 257.0 * 3.0 + 4.0).astype(np.int32)  # RANS#allocateIfNeeded
        blk["out_cap"] = cap
        slot = (cap.astype(np.int64) + 15) // 16 * 16
        blk["out_off"] = np.cumsum(slot) - slot
        assert int(blk["out_off"][-1] + slot[-1]) + 16 <= self.d_out.numel()
        if timed:
            self.ev[0].record()
        self.eng.encode_dev(self.writer.arena, self.d_out, blk, self.nb, self.d_len, self.d_pos)
        if timed:
            self.ev[1].record()
        self.eng.gather_dev(self.d_out, self.d_pos, self.d_len, self.nb, int(cap.max()), self.d_pack, self.d_pack_off)
        self.raw_bytes = int(blk["in_len"].sum())
        self.last = (feats, mats, scores, force, status, merged, off, ln)
        if timed:
            torch.cuda.synchronize()
            self.enc_ms.append(self.ev[0].elapsed_time(self.ev[1]))

    def verify(self, rs, ref_np):
        """Container 0 against the oracle: merged features, every series of the writer, every rANS stream."""
        from oracle import orc
        m = min(READS_PER_CONTAINER, rs["n"])
        co, nc = rs["cigar_off"], rs["n_cigar"]
        cig = [list(rs["cigar"][co[r]:co[r] + nc[r]]) for r in range(m)]
        L = int(rs["read_len"][0])
        bs = [rs["bases"][r * L:(r + 1) * L].tobytes() for r in range(m)]
        qs = [rs["quals"][r * L:(r + 1) * L].tobytes() for r in range(m)]
        o = orc.Reads(rs["aln_start"][:m], cig, bs, quals=qs, mapq=rs["mapq"][:m])
        wf = orc.extract_features(o, ref_np)
        wm = orc.substitution_matrix(wf["hist"])
        orc.assign_codes(wf, wm)
        pols = orc.parse_policies(self.spec)
        wpos = orc.tracks_windows(o, pols, ref_np.size)
        ws, wforce, wst = orc.lossy_quals(o, ref_np, pols, 1000000, wpos)
        merged_w = orc.merge_quality_features(o, ws, wforce, wst, wf)
        feats, mats, scores, force, status, merged, off, ln = self.last
        assert np.array_equal(mats[0], wm)
        gm = merged.to_host()
        k = len(merged_w["op"])
        assert int(gm["off"][m]) == k
        for key in ("op", "pos", "a", "b"):
            assert np.array_equal(gm[key][:k], merged_w[key]), key
        full = {c: np.zeros(m, np.int32) for c in rec.RECORD_COLS}
        full.update(bam_flags=np.zeros(m, np.int32), read_group=np.full(m, -1, np.int32), mate_ref=np.full(m, -1, np.int32),
                    comp_flags=(2 | wforce.astype(np.int32)))
        want = orc.write_series(full, o.read_len, o.aln_start, rs["mapq"][:m], o.bases, ws, o.base_off, merged_w)
        lens, pos = self.d_len.cpu().numpy(), self.d_pos.cpu().numpy()
        for i, (name, od) in enumerate(SERIES):
            sid = _lib.SERIES_ID[name]
            assert self.writer.series(0, sid) == want[sid], name
            got = self.d_out[int(pos[i]):int(pos[i]) + int(lens[i])].cpu().numpy().tobytes()
            assert got == orc.rans(want[sid], od), name
        return "container 0: merged features, the ten rANS series of the writer and their streams byte-identical to the CPU oracle"


def main_config4(a, rank, world, dist, be):
    """bench.py --config 4: the JSON line of configs[4] (same shape as the headline's)."""
    n = a.reads if a.reads != 10_000_000 else 20_000
    L = a.read_len if a.read_len != 150 else 10_000
    assert n % READS_PER_CONTAINER == 0 or n < READS_PER_CONTAINER, "whole containers (CramContext.java:33)"
    n_cont = max(1, n // READS_PER_CONTAINER)
    rng = np.random.default_rng(1 + 10 * rank)
    ref_np = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, max(n * L // 8, 4 * L) + 16 * L)]
    rs = long_read_set(rng, n, L, ref_np)
    ref = torch.as_tensor(np.concatenate([ref_np, np.zeros(16, np.uint8)])).to(be.dev)
    spec = "R8-N40-D8"
    step = StepC5(be.dev, be.eng, ref, rs, spec, n_cont)
    for _ in range(a.warmup):
        step.run(False)
    dt = sharding.timed_region(lambda: step.run(True), a.steps, dist=dist, sync=be.sync)
    lens = step.d_len.cpu().numpy()
    assert (lens >= 0).all()
    comp = int(lens.sum())
    line = None
    if rank == 0:
        parity = step.verify(rs, ref_np)
        enc_s = float(np.mean(step.enc_ms)) / 1e3
        algo = step.raw_bytes + comp
        raw = step.raw_bytes * world
        line = dict(metric="CRAM encode hot path: uncompressed external-block payload MB/s (rANS o0/o1 + feature extraction)",
                    value=raw * a.steps / dt / 1e6, unit="MB/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                    ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8", data="synthetic",
                    config=dict(workload="configs[4]: %d x %d bp reads (3 %% substitutions, an insertion or deletion of 1-3 bases every "
                                         "12.5 bases), lossy model %s, %d container(s) x %d rANS blocks: K3 -> F3 -> K5 -> merge -> "
                                         "series writer -> rANS -> gather" % (n, L, spec, n_cont, len(SERIES)), reads_per_gpu=n,
                                parallelism="containers sharded by index, %d rank(s), no collective" % world),
                    reads_per_s=n * world * a.steps / dt, bases_per_s=n * L * world * a.steps / dt,
                    compression_ratio=step.raw_bytes / max(comp, 1),
                    roofline=dict(bound="hbm", achieved=algo / enc_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=algo / enc_s / 1e9 / HBM_PEAK_GBS, traffic=None, kernel="rans_encode_kernel",
                                  ms_per_launch=enc_s * 1e3, algorithmic_bytes=algo,
                                  note="one 10^8-byte quality block per container is ONE serial rANS chain of 2.5 x 10^7 steps: "
                                       "latency bound by construction (SURVEY.md section 7)"),
                    parity=parity)
        print(json.dumps(line), flush=True)
    be.close()
    if world > 1:
        dist.destroy_process_group()
    return line


PMC_ROUND = "r03"  # the round whose committed --pmc passes describe the kernels as they are now


def pmc_traffic(a):
    """HBM bytes per rans_encode_kernel launch from the committed rocprofv3 --pmc passes of THIS round
    (profiles/<PMC_ROUND>_pmc_{FETCH,WRITE}_SIZE.csv, taken with this same command at the default sizes): 2 x FETCH_SIZE
    (gfx950 tallies the 128-byte requests of 16-byte streaming loads at 64 bytes, MI355X_MICROARCH.md section HBM) +
    WRITE_SIZE, both in KiB.  A stored counter, not one of this run: counters cannot be collected from inside the timed
    process; the line names the files (roofline.traffic_source).  None at other sizes or when the passes are missing."""
    if (a.reads, a.ref_len, a.read_len) != (10_000_000, 64_000_000, 150):
        return None
    import csv
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = [os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (PMC_ROUND, c))]
        if not os.path.exists(files[0]):
            return None
        with open(files[-1]) as f:
            rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
        for r in rows[1:]:
            if "rans_encode_kernel" in r[0]:
                vals[c] = float(r[2]) * 1024.0
    if len(vals) != 2:
        return None
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]


class GpuBackend:
    """What main() needs from the device side: the real thing on one MI355X.  tests/test_bench_ranks.py passes a CPU
    stand-in with the same methods to run main()'s rank logic under gloo."""
    dist_backend = "nccl"

    def __init__(self, local):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
        torch.cuda.set_device(local)
        self.dev = "cuda:%d" % local
        self.eng = xc.Engine(local)

    def workload(self, a, seed_shift):
        return build_workload(self.dev, a.reads, a.ref_len, a.read_len, seed_shift,
                              lambda reads, ref: gpu_insert_bases(self.eng, reads, ref, a.reads))

    def step(self, w):
        return Step(self.dev, self.eng, w)

    def decode_step(self, w, step):
        return DecodeStep(self.dev, self.eng, w, step)

    def sync(self):
        torch.cuda.synchronize()

    def pcie(self, w):
        return pcie_inclusive(self.eng, w)

    def pcie_coalesced(self, w):
        return pcie_coalesced(self.eng, w, threads=(64,), tickets=(64, 128))

    def close(self):
        self.eng.close()


def main(argv=None, backend_factory=GpuBackend):
    """Returns the JSON line's dict on rank 0 (None elsewhere) after printing it."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--ref-len", type=int, default=64_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--no-full", action="store_true", help="skip the encode_full / decode_full regions")
    ap.add_argument("--no-bgzf", action="store_true", help="skip the BGZF image of the BAM stripe (2.7 GB through the gzip kernel)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 4),
                    help="1: the headline (BASELINE.json configs[1] + the configs[2] decode leg); 4: configs[4], long reads + lossy model")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend_factory.dist_backend)
    be = backend_factory(local)
    if a.config == 4:
        return main_config4(a, rank, world, dist, be)

    # every rank codes its own containers (seed shifted by rank): weak scaling, no collective on the data path
    w = be.workload(a, 10 * rank)
    step = be.step(w)
    for _ in range(a.warmup):
        step.run(False)
    # barrier + synchronize on both sides of exactly `steps` steps, MAX over ranks (cramtools_amd/sharding.py)
    dt = sharding.timed_region(lambda: step.run(True), a.steps, dist=dist, sync=be.sync)

    # decode leg (configs[2]) on every rank, outside the timed region of the headline metric, timed the same way
    dec, dec_parity, dt_dec = None, None, None
    if not a.no_decode:
        dec = be.decode_step(w, step)
        dec.prepare()
        dec.run()
        dec.k2_ms, dec.ms, dec.gz_ms = [], [], []
        dt_dec = sharding.timed_region(dec.run, a.steps, dist=dist, sync=be.sync)
        dec_parity = dec.verify()
        if getattr(dec, "reader", None) is not None:  # container assembly + BAM records on the restored reads
            dec.assemble(False)
            for _ in range(a.steps):
                dec.assemble(True)
            dec_parity += dec.verify_assembly()

    # parity verdict on this rank's blocks: no negative lengths, and the sampled containers equal the CPU oracle
    lens = step.d_len.cpu().numpy()
    assert (lens >= 0).all(), "a block failed to encode: %r" % (lens[lens < 0][:8],)
    if getattr(step, "writer", None) is not None:
        step.writer.lengths()  # raises what the writer reported
        assert int(step.d_blk_status.item()) == 0, "the block table made on the device: status %d" % int(step.d_blk_status.item())
    comp_bytes = int(lens[lens > 0].sum())
    extra = {}
    line = None
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:  # the CPU leg and its block-by-block parity run at N = 1 only
            base, (out, ooff, olen, n_c) = cpu_baseline(w)
            extra["cpu_baseline"] = base
            ns = len(SERIES)
            pos = step.d_pos.cpu().numpy()
            for i in range(n_c * ns):
                got = step.d_out[int(pos[i]):int(pos[i]) + int(lens[i])].cpu().numpy()
                assert int(lens[i]) == int(olen[i]) and np.array_equal(got, out[ooff[i]:ooff[i] + olen[i]]), \
                    "block %d differs from the oracle" % i
            extra["parity"] = "first %d blocks byte-identical to the CPU oracle" % (n_c * ns)
        enc_s = float(np.mean(step.enc_ms)) / 1e3
        algo = w["raw_bytes"] + comp_bytes  # n_raw + n_comp per encoded block
        ach = algo / enc_s / 1e9
        extra["roofline"] = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                 traffic=pmc_traffic(a),
                                 traffic_source="profiles/%s_pmc_{FETCH,WRITE}_SIZE.csv (stored rocprofv3 --pmc passes)" % PMC_ROUND,
                                 kernel="rans_encode_kernel", ms_per_launch=enc_s * 1e3,
                                 algorithmic_bytes=algo)
        if dec is not None:
            k2_s, dec_s = float(np.mean(dec.k2_ms)) / 1e3, dt_dec / a.steps
            extra["decode"] = dict(workload="configs[2]: rANS decode of the same %d blocks + gunzip of the GZIP series + the "
                                            "data-series reader + restoreReadBases / restoreQualityScores of the %d reads per rank (timed outside the K steps, barrier "
                                            "and MAX over ranks like them)" % (w["n_blocks"], a.reads),
                                   value=w["raw_bytes"] * world / dec_s / 1e6, unit="MB/s", reads_per_s=a.reads * world / dec_s,
                                   ms_per_step=dec_s * 1e3, rans_decode_kernel_ms=k2_s * 1e3,
                                   gunzip=(dict(ms_from_leg_start=float(np.mean(dec.gz_ms)) if dec.gz_ms else None, streams=dec.n_gz, gz_bytes=dec.gz_bytes[0],
                                                raw_bytes=dec.gz_bytes[1],
                                                note="the series the reference stores as GZIP blocks, inflated on the device "
                                                     "inside the leg (zlib level 5 made them at set-up)")
                                           if getattr(dec, "n_gz", 0) else None),
                                   rans_decode_GBps=algo / k2_s / 1e9, rans_decode_frac_of_hbm=algo / k2_s / 1e9 / HBM_PEAK_GBS,
                                   parity=dec_parity)
            if hasattr(dec, "gzip_series"):
                gw = dec.gzip_series()
                if gw is not None:
                    cut = dec.gzip_series(piece=8192)
                    gw.update(ms_cut_8k=cut["ms"], members_cut_8k=cut["members"], gz_bytes_cut_8k=cut["gz_bytes"],
                              note="ExternalCompression#gzip of the same series on the device (cramgpu_gzip_dev), outside every timed "
                                   "region; every member inflated again by the gunzip kernel and compared with the series; "
                                   "*_cut_8k: series longer than 8 KiB written as several members back to back")
                extra["decode"]["gzip_writer"] = gw
            if getattr(dec, "bam_ms", None):
                bam_s = float(np.mean(dec.bam_ms)) / 1e3
                extra["decode"]["assembly"] = dict(
                    workload="SURVEY 8(f) rows 4 + 3 on the decoded reads: CIGARs from features, mate restoration, one BAM record "
                             "per read (the uncompressed stripe handed to BGZF); outside the decode leg's time",
                    ms=bam_s * 1e3, bam_bytes=dec.bam_bytes, GBps_written=dec.bam_bytes / bam_s / 1e9,
                    reads_per_s=a.reads / bam_s,
                    bam_decode_ms=getattr(dec, "bam_decode_ms", None),  # the stripe back into columns (BAMRecordCodec#decode)
                    bam_index_ms=getattr(dec, "bam_index_ms", None))    # its record boundaries found on the device
                if not a.no_bgzf and hasattr(dec, "bam_bytes"):
                    extra["decode"]["assembly"]["bgzf"] = bgzf_image(dec)
        if not a.no_full and world == 1 and dec is not None and getattr(dec, "reader", None) is not None and hasattr(step, "run_full"):
            step.prepare_full()
            step.run_full(False)
            for _ in range(a.steps):
                step.run_full(True)
            ef = float(np.mean(step.full_ms))
            extra["encode_full"] = dict(
                workload="one region per step, CramSerilization.java:108-200 per container: K3 -> F3 -> setMateInfo (unique read "
                         "names: every record detached) -> W -> Z (the %d non-rANS series as gzip members, pieces of 8 KiB) -> K1 -> "
                         "gather; slice reference MD5 and ContentDigests BD / SD on a second stream beside them" % len(step.rest),
                ms=ef, value=w["raw_bytes"] / ef / 1e3, unit="MB/s", reads_per_s=a.reads / ef * 1e3, **step.verify_full())
            dec.run_full(False)
            for _ in range(a.steps):
                dec.run_full(True)
            df = float(np.mean(dec.full_ms))
            extra["decode_full"] = dict(
                workload="one region per step, CramToBam_OBA_Function.java:53-83 per container: K2 beside G -> R -> "
                         "restoreReadBases / restoreQualityScores -> CIGARs, mate restoration, BAM records -> the stripe as BGZF "
                         "members (Z)", ms=df, value=w["raw_bytes"] / df / 1e3, unit="MB/s", reads_per_s=a.reads / df * 1e3,
                **dec.verify_full())
            dec.run_ingest(False)
            for _ in range(a.steps):
                dec.run_ingest(True)
            gi = float(np.mean(dec.ingest_ms))
            extra["ingest_full"] = dict(
                workload="one region per step, the way in of a BAM -> CRAM conversion (Bam_OBA_Supplier.java:45-82, BAMRecordCodec"
                         "#decode): the %d BGZF members decode_full wrote -> G -> record boundaries on the device -> BAM records "
                         "into columns (what K3 takes)" % dec.bgzf_n, ms=gi, bam_bytes=dec.bam_bytes,
                GBps_inflated=dec.bam_bytes / gi / 1e6, reads_per_s=a.reads / gi * 1e3, parity=dec.verify_ingest())
        if not a.no_pcie and world == 1:
            extra["pcie_inclusive"] = be.pcie(w)
            extra["pcie_inclusive"]["hw_queues"] = HW_QUEUES
            if hasattr(be, "pcie_coalesced"):
                extra["pcie_inclusive"]["one_container_per_submission"] = be.pcie_coalesced(w)
        total_raw = w["raw_bytes"] * world
        line = dict(metric="CRAM encode hot path: uncompressed external-block payload MB/s (rANS o0/o1 + feature extraction)",
                    value=total_raw * a.steps / dt / 1e6, unit="MB/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                    ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8",
                    data="synthetic",
                    config=dict(workload="configs[1]: %d x %d bp reads vs %d bp reference, %d containers x %d rANS blocks, lossless"
                                % (a.reads, a.read_len, a.ref_len, w["n_cont"], len(SERIES)), reads_per_gpu=a.reads,
                                parallelism="containers sharded by index, %d rank(s), no collective" % world),
                    reads_per_s=a.reads * world * a.steps / dt, compression_ratio=w["raw_bytes"] / max(comp_bytes, 1))
        line.update(extra)
        print(json.dumps(line), flush=True)
    be.close()
    if world > 1:
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
