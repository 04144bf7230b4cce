"""N > 1 bookkeeping of the container-sharded path, on the CPU: world-size-2 gloo run of the sharding / ordering /
timing helpers bench.py uses (the codec itself is stood in for by the CPU oracle here — tests may use it)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from cramtools_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_and_ordering_buffer():
    assert sharding.container_shard(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((sharding.container_shard(11, r, 3) for r in range(3)), [])) == list(range(11))
    ob = sharding.OrderingBuffer()
    assert ob.put(2, "c") == [] and ob.put(1, "b") == []
    assert ob.put(0, "a") == ["a", "b", "c"]
    assert ob.put(3, "d") == ["d"] and ob.pending() == 0


class _Eng:
    def __init__(self, tag):
        self.tag = tag

    def run(self, job):
        return (self.tag, job * 2)


def test_container_queue_orders_results():
    q = sharding.ContainerQueue([_Eng("g0"), _Eng("g1"), _Eng("g2")])
    for k in range(20):
        q.submit(k, k + 100)
    out = q.run()
    assert [o for o, _ in out] == list(range(20))
    assert all(res == ("g%d" % (o % 3), (o + 100) * 2) for o, res in out)


class _SlowEng:
    def __init__(self, tag, seed):
        import random
        self.tag, self.rng = tag, random.Random(seed)

    def run(self, job):
        import time
        time.sleep(self.rng.random() * 0.004)
        return (self.tag, job * 2)


def test_container_queue_orders_results_under_skew():
    """Workers that finish at random times still hand their results on in ordinal order (the release list
    used to be appended outside the ordering lock)."""
    for seed in range(5):
        q = sharding.ContainerQueue([_SlowEng("g%d" % g, 10 * seed + g) for g in range(4)])
        for k in range(60):
            q.submit(k, k)
        out = q.run()
        assert [o for o, _ in out] == list(range(60))


def test_engine_coder_adapter():
    class FakeEngine:
        def compress_many(self, datas, orders):
            return [bytes([o]) + d for d, o in zip(datas, orders)]

        def uncompress_many(self, streams):
            return [s[1:] for s in streams]
    q = sharding.ContainerQueue([sharding.EngineCoder(FakeEngine()) for _ in range(2)])
    for k in range(5):
        q.submit(k, ([b"abc%d" % k], [k & 1]))
    out = q.run()
    assert [r for _, r in out] == [[bytes([k & 1]) + b"abc%d" % k] for k in range(5)]
    q = sharding.ContainerQueue([sharding.EngineCoder(FakeEngine())])
    q.submit(0, (out[0][1], None))
    assert q.run() == [(0, [b"abc0"])]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cramtools_amd import sharding as sh, synth
    from oracle import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_cont = 7
    mine = sh.container_shard(n_cont, rank, world)
    res = {}

    def step():
        for k in mine:
            res[k] = orc.rans(synth.markov_quals(3000, 10 + k).tobytes(), 1)
    dt = sh.timed_region(step, 2, dist=dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        ob = sh.OrderingBuffer()
        ordered = []
        for part in gathered:
            for k, v in part.items():
                ordered += ob.put(k, (k, v))
        q.put((dt, [k for k, _ in ordered], [v for _, v in ordered]))
    dist.destroy_process_group()


def test_two_rank_gloo_run_matches_single_process(orc):
    from cramtools_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    dt, order, streams = q.get(timeout=120)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert dt > 0 and order == list(range(7))
    want = [orc.rans(synth.markov_quals(3000, 10 + k).tobytes(), 1) for k in range(7)]
    assert streams == want
