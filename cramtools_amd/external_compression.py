"""Host-side mirror of the reference's block-codec plugin interface, backed by libcramgpu (HIP, gfx950 only).

Reference interface (SURVEY.md §8(b) level 0/1):
  htsjdk!io/ExternalCompression#rans(byte[], int|ORDER) L89-90,101-102   -> rans(data, order)
  htsjdk!io/ExternalCompression#unrans(byte[]) L112-113                  -> unrans(data)
  htsjdk!encoding/ExternalCompressor#createRANS(ORDER) L76, $5#compress  -> ExternalCompressor.createRANS(order).compress
  htsjdk!structure/BlockCompressionMethod                                -> ExternalCompressor.getMethod() == 4 (RANS)
Same argument meaning and error behaviour: empty in -> empty out, < 4 bytes -> order 0, errors raise RuntimeError
subclasses carrying the reference's messages.  `Engine` is the batch (whole-container) form over device arenas.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Block, CramGpuError

ORDER_ZERO, ORDER_ONE = 0, 1
RANS_METHOD = 4  # BlockCompressionMethod.RANS ordinal


def rans_bound(n):
    return int(_lib.lib().cramgpu_rans_bound(int(n)))


def rans(data, order):
    """ExternalCompression.rans(byte[] data, int order) -> bytes."""
    if order not in (0, 1):
        raise CramGpuError(_lib.E_ORDER, arg=order)  # RANS.ORDER.fromInt: "Unknown rANS order: X"
    d = np.frombuffer(bytes(data), np.uint8)
    if d.size == 0:
        return b""
    cap = rans_bound(d.size)
    out = np.empty(cap, np.uint8)
    r = _lib.lib().cramgpu_rans_compress(d.ctypes.data, d.size, order, out.ctypes.data, cap)
    _lib.check(r)
    return out[:r].tobytes()  # ExternalCompression#toByteArray L146-150: exact-size copy


def unrans(data):
    """ExternalCompression.unrans(byte[] data) -> bytes."""
    d = np.frombuffer(bytes(data), np.uint8)
    if d.size == 0:
        return b""
    raw = _lib.check(_lib.lib().cramgpu_rans_raw_size(d.ctypes.data, d.size))
    out = np.empty(max(raw, 1), np.uint8)
    r = _lib.lib().cramgpu_rans_uncompress(d.ctypes.data, d.size, out.ctypes.data, raw)
    if r == _lib.E_OUT_TOO_SMALL:  # RANS#uncompress: "Output buffer too small to fit N bytes."
        raise CramGpuError(r, arg=raw)
    _lib.check(r)
    return out[:r].tobytes()


class ExternalCompressor:
    """htsjdk!encoding/ExternalCompressor: the per-data-series plugin object (rANS flavours only)."""

    def __init__(self, order):
        self.order = order

    @staticmethod
    def createRANS(order):
        return ExternalCompressor(order)

    def getMethod(self):
        return RANS_METHOD

    def compress(self, data):
        return rans(data, self.order)


def _align(v, a=_lib.ARENA_ALIGN):
    return (int(v) + a - 1) // a * a


class Engine:
    """One context (= one GPU).  Batch encode/decode over device arenas held as torch uint8 tensors."""

    def __init__(self, device_index=0):
        self._h = _lib.lib().cramgpu_open(int(device_index))
        if not self._h:
            raise CramGpuError(_lib.E_NO_DEVICE, _lib.last_error())
        self.device_index = int(device_index)

    def close(self):
        if self._h:
            _lib.lib().cramgpu_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self):
        a, b = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().cramgpu_info(self._h, C.byref(a), C.byref(b)))
        return {"n_cu": a.value, "n_waves": b.value}

    # ---- layout helpers (host) ----
    @staticmethod
    def plan_encode(lengths, orders):
        """Descriptor table for raw blocks packed back to back (16-byte aligned) -> (Block array, in_bytes, out_bytes)."""
        n = len(lengths)
        blocks = (Block * max(n, 1))()
        io = oo = 0
        for i in range(n):
            ln = int(lengths[i])
            cap = rans_bound(ln)
            blocks[i] = Block(io, oo, ln, cap, int(orders[i]), 0)
            io += _align(ln)
            oo += _align(cap)
        return blocks, io + 16, oo + 16

    @staticmethod
    def plan_decode(stream_lengths, raw_sizes):
        n = len(stream_lengths)
        blocks = (Block * max(n, 1))()
        io = oo = 0
        for i in range(n):
            blocks[i] = Block(io, oo, int(stream_lengths[i]), int(raw_sizes[i]), 0, 0)
            io += _align(stream_lengths[i])
            oo += _align(raw_sizes[i])
        return blocks, io + 16, oo + 16

    # ---- device calls (torch tensors in, stream-ordered on torch's current stream) ----
    def encode_dev(self, d_in, d_out, blocks, n, d_out_len, d_out_pos, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        # blocks: a ctypes array of Block or a numpy structured array of the same layout
        pb = blocks.ctypes.data if hasattr(blocks, "ctypes") else C.addressof(blocks)
        p_in = d_in if isinstance(d_in, int) else d_in.data_ptr()  # an address: blocks spread over several device buffers
        _lib.check(_lib.lib().cramgpu_rans_encode_dev(self._h, p_in, d_out.data_ptr(), pb, n,
                                                      d_out_len.data_ptr(), d_out_pos.data_ptr(), s))

    def encode_series_dev(self, base, in_shift, d_series_off, d_series_len, d_w_status, n_cont, ids, orders, d_out, d_blocks,
                          d_out_len, d_out_pos, d_status, stream=None):
        """The encoder straight behind the data-series writer (cramgpu_rans_encode_series_dev): block c * len(ids) + k is
        series ids[k] of container c; the block table (d_blocks, a DEVICE uint8 tensor of 32 bytes per block) and its
        longest-first order are made on the device, nothing comes back to the host before the launch."""
        import numpy as np
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        ids = np.ascontiguousarray(ids, np.int32)
        orders = np.ascontiguousarray(orders, np.int32)
        assert ids.size == orders.size and d_blocks.numel() >= 32 * n_cont * ids.size
        p_base = base if isinstance(base, int) else base.data_ptr()
        _lib.check(_lib.lib().cramgpu_rans_encode_series_dev(
            self._h, p_base, int(in_shift), d_series_off.data_ptr(), d_series_len.data_ptr(),
            d_w_status.data_ptr() if d_w_status is not None else None, int(n_cont), ids.ctypes.data, orders.ctypes.data,
            int(ids.size), d_out.data_ptr(), int(d_out.numel()), d_blocks.data_ptr(), d_out_len.data_ptr(),
            d_out_pos.data_ptr(), d_status.data_ptr(), s))

    def decode_dev(self, d_in, d_out, blocks, n, d_out_len, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cramgpu_rans_decode_dev(self._h, d_in.data_ptr(), d_out.data_ptr(), C.addressof(blocks), n,
                                                      d_out_len.data_ptr(), s))

    def gunzip_dev(self, d_in, d_out, d_blocks, n, d_out_len, stream=None):
        """ExternalCompression#gunzip for n streams at once (cramgpu_gunzip_dev): d_blocks = DEVICE tensor holding n
        cramgpu_block descriptors (BLOCK dtype bytes)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cramgpu_gunzip_dev(self._h, d_in.data_ptr(), d_out.data_ptr(), d_blocks.data_ptr(), n,
                                                 d_out_len.data_ptr(), s))

    def gunzip_many(self, datas, raw_sizes, device="cuda:0"):
        """[gzip bytes], [raw size] -> [bytes]; raises CramGpuError for the first stream that fails."""
        import torch
        n = len(datas)
        if n == 0:
            return []
        desc = (_lib.Block * n)()
        in_off = out_off = 0
        for i, d in enumerate(datas):
            desc[i] = _lib.Block(in_off, out_off, len(d), int(raw_sizes[i]), 0, 0)
            in_off += len(d)
            out_off += int(raw_sizes[i])
        d_in = torch.as_tensor(np.frombuffer(b"".join(bytes(d) for d in datas) + bytes(16), np.uint8).copy()).to(device)
        d_out = torch.zeros(out_off + 16, dtype=torch.uint8, device=device)
        d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(device)
        d_len = torch.zeros(n, dtype=torch.int32, device=device)
        self.gunzip_dev(d_in, d_out, d_desc, n, d_len)
        lens = d_len.cpu().numpy()
        out = d_out.cpu().numpy()
        res = []
        for i in range(n):
            if lens[i] < 0:
                raise CramGpuError(int(lens[i]), "gzip stream %d" % i)
            res.append(out[desc[i].out_off:desc[i].out_off + int(lens[i])].tobytes())
        return res

    def gzip_dev(self, d_in, d_out, d_blocks, n, d_out_len, bgzf=False, stream=None):
        """ExternalCompression#gzip for n streams at once (cramgpu_gzip_dev): one gzip member per descriptor, BGZF
        members (with the 'BC' size field) when bgzf is set."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cramgpu_gzip_dev(self._h, d_in.data_ptr(), d_out.data_ptr(), d_blocks.data_ptr(), n,
                                               d_out_len.data_ptr(), 1 if bgzf else 0, s))

    @staticmethod
    def gzip_bound(n):
        return int(_lib.check(_lib.lib().cramgpu_gzip_bound(int(n))))

    def gzip_many(self, datas, bgzf=False, device="cuda:0", caps=None):
        """[bytes] -> [gzip member]; raises CramGpuError for the first stream that fails (a slot too small, a BGZF
        member beyond 64 KiB)."""
        import torch
        n = len(datas)
        if n == 0:
            return []
        desc = (_lib.Block * n)()
        in_off = out_off = 0
        for i, d in enumerate(datas):
            cap = self.gzip_bound(len(d)) if caps is None else int(caps[i])
            desc[i] = _lib.Block(in_off, out_off, len(d), cap, 0, 0)
            in_off += len(d)
            out_off += cap
        d_in = torch.as_tensor(np.frombuffer(b"".join(bytes(d) for d in datas) + bytes(16), np.uint8).copy()).to(device)
        d_out = torch.zeros(out_off + 16, dtype=torch.uint8, device=device)
        d_desc = torch.as_tensor(np.frombuffer(bytes(desc), np.uint8).copy()).to(device)
        d_len = torch.zeros(n, dtype=torch.int32, device=device)
        self.gzip_dev(d_in, d_out, d_desc, n, d_len, bgzf)
        lens = d_len.cpu().numpy()
        out = d_out.cpu().numpy()
        res = []
        for i in range(n):
            if lens[i] < 0:
                raise CramGpuError(int(lens[i]), "gzip stream %d" % i)
            res.append(out[desc[i].out_off:desc[i].out_off + int(lens[i])].tobytes())
        return res

    def gather_dev(self, d_src, d_pos, d_len, n, max_len, d_dst, d_dst_off, align=16, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cramgpu_gather_dev(self._h, d_src.data_ptr(), d_pos.data_ptr(), d_len.data_ptr(), n,
                                                 int(max_len), align, d_dst.data_ptr(), d_dst.numel(),
                                                 d_dst_off.data_ptr(), s))

    # ---- host-arena batches (numpy) ----
    def compress_many(self, datas, orders):
        """[bytes], [order] -> [bytes] through cramgpu_rans_encode_batch (one H2D, one launch, one D2H)."""
        n = len(datas)
        if n == 0:
            return []
        blocks, in_bytes, out_bytes = self.plan_encode([len(d) for d in datas], orders)
        arena = np.zeros(in_bytes, np.uint8)
        for i, d in enumerate(datas):
            arena[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(bytes(d), np.uint8)
        out = np.zeros(out_bytes, np.uint8)
        lens = np.zeros(n, np.int32)
        _lib.check(_lib.lib().cramgpu_rans_encode_batch(self._h, arena.ctypes.data, in_bytes, out.ctypes.data, out_bytes,
                                                        C.addressof(blocks), n, lens.ctypes.data))
        res = []
        for i in range(n):
            if lens[i] < 0:
                raise CramGpuError(int(lens[i]), "block %d" % i)
            res.append(out[blocks[i].out_off:blocks[i].out_off + lens[i]].tobytes())
        return res

    # ---- asynchronous host-arena batches (cramgpu_rans_submit / cramgpu_job_wait) ----
    def submit_compress(self, datas, orders):
        """Queues one batch (e.g. the external blocks of one container) and returns a job; `job.wait()` -> [bytes].
        The arenas are pinned memory of the library (cramgpu_host_alloc), what the JNI layer hands to Java as direct
        ByteBuffers."""
        return _Job(self, datas, orders, decode=False)

    def submit_uncompress(self, streams):
        return _Job(self, streams, None, decode=True)

    def uncompress_many(self, streams):
        n = len(streams)
        if n == 0:
            return []
        raws = []
        for s in streams:
            b = np.frombuffer(bytes(s), np.uint8)
            raws.append(0 if b.size == 0 else _lib.check(_lib.lib().cramgpu_rans_raw_size(b.ctypes.data, b.size)))
        blocks, in_bytes, out_bytes = self.plan_decode([len(s) for s in streams], raws)
        arena = np.zeros(in_bytes, np.uint8)
        for i, s in enumerate(streams):
            arena[blocks[i].in_off:blocks[i].in_off + len(s)] = np.frombuffer(bytes(s), np.uint8)
        out = np.zeros(out_bytes, np.uint8)
        lens = np.zeros(n, np.int32)
        _lib.check(_lib.lib().cramgpu_rans_decode_batch(self._h, arena.ctypes.data, in_bytes, out.ctypes.data, out_bytes,
                                                        C.addressof(blocks), n, lens.ctypes.data))
        res = []
        for i in range(n):
            if lens[i] < 0:
                raise CramGpuError(int(lens[i]), "block %d" % i)
            res.append(out[blocks[i].out_off:blocks[i].out_off + lens[i]].tobytes())
        return res


class _Job:
    """One outstanding cramgpu_rans_submit: owns its pinned arenas until wait()."""

    def __init__(self, eng, items, orders, decode):
        L = _lib.lib()
        n = len(items)
        if decode:
            raws = []
            for s in items:
                b = np.frombuffer(bytes(s), np.uint8)
                raws.append(0 if b.size == 0 else _lib.check(L.cramgpu_rans_raw_size(b.ctypes.data, b.size)))
            self.blocks, self.in_bytes, self.out_bytes = Engine.plan_decode([len(s) for s in items], raws)
        else:
            self.blocks, self.in_bytes, self.out_bytes = Engine.plan_encode([len(d) for d in items], orders)
        self.n, self.eng = n, eng
        self.h_in = L.cramgpu_host_alloc(self.in_bytes)
        self.h_out = L.cramgpu_host_alloc(self.out_bytes)
        self.h_len = L.cramgpu_host_alloc(4 * max(n, 1))
        if not (self.h_in and self.h_out and self.h_len):
            raise CramGpuError(_lib.E_HIP, "cramgpu_host_alloc failed")
        arena = np.ctypeslib.as_array(C.cast(self.h_in, C.POINTER(C.c_uint8)), shape=(self.in_bytes,))
        arena[:] = 0
        for i, d in enumerate(items):
            arena[self.blocks[i].in_off:self.blocks[i].in_off + len(d)] = np.frombuffer(bytes(d), np.uint8)
        self.job = L.cramgpu_rans_submit(eng.handle, 1 if decode else 0, self.h_in, self.in_bytes, self.h_out, self.out_bytes,
                                         C.addressof(self.blocks), n, self.h_len)
        if not self.job:
            self._free()
            raise CramGpuError(_lib.E_ARG, _lib.last_error())

    def _free(self):
        L = _lib.lib()
        for name in ("h_in", "h_out", "h_len"):
            p = getattr(self, name, None)
            if p:
                L.cramgpu_host_free(p)
                setattr(self, name, None)

    def done(self):
        return _lib.check(_lib.lib().cramgpu_job_poll(self.job)) == 1

    def wait(self):
        try:
            _lib.check(_lib.lib().cramgpu_job_wait(self.job))
            out = np.ctypeslib.as_array(C.cast(self.h_out, C.POINTER(C.c_uint8)), shape=(self.out_bytes,))
            lens = np.ctypeslib.as_array(C.cast(self.h_len, C.POINTER(C.c_int32)), shape=(max(self.n, 1),))
            res = []
            for i in range(self.n):
                if lens[i] < 0:
                    raise CramGpuError(int(lens[i]), "block %d" % i)
                res.append(out[self.blocks[i].out_off:self.blocks[i].out_off + lens[i]].tobytes())
            return res
        finally:
            self._free()


class Coalescer:
    """The coalescing queue of one Engine (cramgpu_coalescer_*): the form for hosts that keep the reference's thread
    model, one container per worker thread (paralell/TransformerJob.java:29-41).  `rans(datas, orders)` /
    `unrans(streams)` are blocking and may be called from any number of threads at once; the library turns the calls
    that are waiting at the same moment into ONE launch."""

    def __init__(self, eng, max_batches=0):
        self.eng = eng
        self._q = _lib.lib().cramgpu_coalescer_open(eng.handle, max_batches)
        if not self._q:
            raise CramGpuError(_lib.E_ARG, _lib.last_error())

    @property
    def handle(self):
        return self._q

    def close(self):
        if self._q:
            _lib.lib().cramgpu_coalescer_close(self._q)
            self._q = None

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cramgpu_coalesce_stats(self._q, C.byref(a), C.byref(b), C.byref(c)))
        return dict(batches=a.value, submissions=b.value, largest_batch=c.value)

    def _run(self, items, orders, decode):
        L = _lib.lib()
        n = len(items)
        if n == 0:
            return []
        if decode:
            raws = []
            for s in items:
                b = np.frombuffer(bytes(s), np.uint8)
                raws.append(0 if b.size == 0 else _lib.check(L.cramgpu_rans_raw_size(b.ctypes.data, b.size)))
            blocks, in_bytes, out_bytes = Engine.plan_decode([len(s) for s in items], raws)
        else:
            blocks, in_bytes, out_bytes = Engine.plan_encode([len(d) for d in items], orders)
        h_in, h_out, h_len = L.cramgpu_host_alloc(in_bytes), L.cramgpu_host_alloc(max(out_bytes, 16)), L.cramgpu_host_alloc(4 * n)
        try:
            if not (h_in and h_out and h_len):
                raise CramGpuError(_lib.E_HIP, "cramgpu_host_alloc failed")
            arena = np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(in_bytes,))
            arena[:] = 0
            for i, d in enumerate(items):
                arena[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(bytes(d), np.uint8)
            _lib.check(L.cramgpu_rans_coalesced(self._q, 1 if decode else 0, h_in, in_bytes, h_out, out_bytes,
                                                C.addressof(blocks), n, h_len, None))
            out = np.ctypeslib.as_array(C.cast(h_out, C.POINTER(C.c_uint8)), shape=(max(out_bytes, 16),))
            lens = np.ctypeslib.as_array(C.cast(h_len, C.POINTER(C.c_int32)), shape=(n,))
            res = []
            for i in range(n):
                if lens[i] < 0:
                    raise CramGpuError(int(lens[i]), "block %d" % i)
                res.append(out[blocks[i].out_off:blocks[i].out_off + lens[i]].tobytes())
            return res
        finally:
            for p in (h_in, h_out, h_len):
                if p:
                    L.cramgpu_host_free(p)

    def rans(self, datas, orders):
        """ExternalCompression.rans over the blocks of one container, coalesced with the other threads' calls."""
        return self._run(datas, orders, False)

    def unrans(self, streams):
        return self._run(streams, None, True)
