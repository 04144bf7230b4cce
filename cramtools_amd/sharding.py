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
