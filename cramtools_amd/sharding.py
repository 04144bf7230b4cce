"""Container sharding over GPUs / ranks.

Containers (= slices here) are independent, so the multi-GPU form of the path is a host-side work queue: container
`k` goes to GPU `k mod G`, results are put back in container order — the role `OrderingConveyer.java:24-38` and
`CRAMContainerAsynchWriter.java:157-189` play for the reference's thread pool.  No collective touches the data path;
`torch.distributed` is used only for the barrier / max-over-ranks timing of bench.py and for gathering small results.
"""
import heapq
import threading
import time


def container_shard(n_containers, rank, world):
    """Ordinals of the containers rank `rank` of `world` codes: k with k mod world == rank."""
    return list(range(rank, n_containers, world))


class OrderingBuffer:
    """Releases (ordinal, payload) pairs strictly in ordinal order (OrderingConveyer.tryAdvance)."""

    def __init__(self, first=0):
        self._next = first
        self._heap = []
        self._lock = threading.Lock()

    def put(self, ordinal, payload, sink=None):
        """Returns the list of payloads that became releasable, in order.  `sink(list)` is called with that list while
        the buffer's lock is still held, so that concurrent producers hand their releases on in ordinal order too
        (two workers each holding a released batch could otherwise append them in either order)."""
        with self._lock:
            heapq.heappush(self._heap, (ordinal, id(payload), payload))
            out = []
            while self._heap and self._heap[0][0] == self._next:
                out.append(heapq.heappop(self._heap)[2])
                self._next += 1
            if sink is not None and out:
                sink(out)
            return out

    def pending(self):
        with self._lock:
            return len(self._heap)


class ContainerQueue:
    """Hands container jobs to one worker thread per engine (GPU) and yields results in container order.

    `engines` are objects with a `.run(job)` method (an `Engine`-backed coder in production; anything in tests).
    """

    def __init__(self, engines):
        self.engines = list(engines)
        self._jobs = [[] for _ in self.engines]
        self._order = OrderingBuffer()
        self._out = []
        self._err = None

    def submit(self, ordinal, job):
        self._jobs[ordinal % len(self.engines)].append((ordinal, job))

    def _work(self, g):
        try:
            for ordinal, job in self._jobs[g]:
                res = self.engines[g].run(job)
                self._order.put(ordinal, (ordinal, res), sink=self._out.extend)
        except Exception as e:  # CRAMContainerAsynchWriter.exceptionInWorker: remember, rethrow to the caller
            self._err = e

    def run(self):
        """Processes everything submitted; returns [(ordinal, result)] in ordinal order."""
        ths = [threading.Thread(target=self._work, args=(g,)) for g in range(len(self.engines))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if self._err is not None:
            raise self._err
        if self._order.pending():
            raise RuntimeError("container ordinals are not contiguous from 0")
        return self._out


class DevicePool:
    """All GPUs of the node behind one handle (cramgpu_pool_*): container `ordinal` goes to GPU `ordinal mod G` through
    that GPU's coalescing queue, results come back in ordinal order (paralell/OrderingConveyer.java:24-38).  `submit` may be
    called from any thread; `next()` from the writer."""

    def __init__(self, n_devices=0, max_batches=0):
        from . import _lib
        self._p = _lib.lib().cramgpu_pool_open(n_devices, max_batches)
        if not self._p:
            raise _lib.CramGpuError(_lib.E_NO_DEVICE, _lib.last_error())
        self._held = {}

    @property
    def devices(self):
        from . import _lib
        return int(_lib.lib().cramgpu_pool_devices(self._p))

    def submit(self, ordinal, datas, orders):
        """The external blocks of one container (orders=None: streams to decode)."""
        import ctypes as C
        import numpy as np
        from . import _lib
        from .external_compression import Engine
        L = _lib.lib()
        decode = orders is None
        n = len(datas)
        if decode:
            raws = []
            for s in datas:
                b = np.frombuffer(bytes(s), np.uint8)
                raws.append(0 if b.size == 0 else _lib.check(L.cramgpu_rans_raw_size(b.ctypes.data, b.size)))
            blocks, in_bytes, out_bytes = Engine.plan_decode([len(s) for s in datas], raws)
        else:
            blocks, in_bytes, out_bytes = Engine.plan_encode([len(d) for d in datas], orders)
        h_in, h_out, h_len = L.cramgpu_host_alloc(in_bytes), L.cramgpu_host_alloc(max(out_bytes, 16)), L.cramgpu_host_alloc(4 * max(n, 1))
        if not (h_in and h_out and h_len):
            raise _lib.CramGpuError(_lib.E_HIP, "cramgpu_host_alloc failed")
        arena = np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(in_bytes,))
        arena[:] = 0
        for i, d in enumerate(datas):
            arena[blocks[i].in_off:blocks[i].in_off + len(d)] = np.frombuffer(bytes(d), np.uint8)
        self._held[ordinal] = (blocks, n, out_bytes, h_in, h_out, h_len)
        _lib.check(L.cramgpu_pool_submit(self._p, ordinal, 1 if decode else 0, h_in, in_bytes, h_out, out_bytes, C.addressof(blocks), n,
                                         h_len, None))

    def next(self):
        """-> (ordinal, [bytes]) of the next container in ordinal order; blocks until it is done."""
        import ctypes as C
        import numpy as np
        from . import _lib
        L = _lib.lib()
        k = C.c_int64(-1)
        rc = L.cramgpu_pool_next(self._p, C.byref(k))
        blocks, n, out_bytes, h_in, h_out, h_len = self._held.pop(k.value)
        try:
            _lib.check(rc)
            out = np.ctypeslib.as_array(C.cast(h_out, C.POINTER(C.c_uint8)), shape=(max(out_bytes, 16),))
            lens = np.ctypeslib.as_array(C.cast(h_len, C.POINTER(C.c_int32)), shape=(max(n, 1),))
            res = []
            for i in range(n):
                if lens[i] < 0:
                    raise _lib.CramGpuError(int(lens[i]), "container %d block %d" % (k.value, i))
                res.append(out[blocks[i].out_off:blocks[i].out_off + lens[i]].tobytes())
            return k.value, res
        finally:
            for p in (h_in, h_out, h_len):
                L.cramgpu_host_free(p)

    def close(self):
        from . import _lib
        if self._p:
            _lib.lib().cramgpu_pool_close(self._p)
            self._p = None


class EngineCoder:
    """`.run(job)` adapter of one `external_compression.Engine` (= one GPU) for `ContainerQueue`: a job is the external
    blocks of one container, `(datas, orders)` to encode or `(streams, None)` to decode, as
    `BamToCram_OBA_Function.apply` / `CramToBam_OBA_Function.apply` see them (paralell/*.java)."""

    def __init__(self, engine):
        self.engine = engine

    def run(self, job):
        items, orders = job
        if orders is None:
            return self.engine.uncompress_many(items)
        return self.engine.compress_many(items, orders)


def timed_region(fn, steps, dist=None, sync=None):
    """bench.py's timing contract: barrier + sync, `steps` calls of fn, sync + barrier, MAX over ranks (seconds)."""
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if sync:
        sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt
