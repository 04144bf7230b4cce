"""bench.py's own rank logic on the CPU: main(argv, backend_factory=...) with a stand-in for the
device side (the CPU oracle codes the blocks — tests may use it), once in one process and once as a world-size-2 gloo
job: per-rank seed shift, barrier + MAX-over-ranks timing, whole-job aggregation and the JSON line's fields."""
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _StubStep:
    def __init__(self, w):
        self.w, self.enc_ms = w, []

    def run(self, timed):
        from oracle import orc
        w = self.w
        t0 = time.perf_counter()
        outs = []
        for b in w["blocks"]:
            outs.append(orc.rans(w["d_in"][b.in_off:b.in_off + b.in_len].numpy().tobytes(), b.order))
        time.sleep(w["delay"])
        lens = np.array([len(o) for o in outs], np.int32)
        pos = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        self.d_len, self.d_pos = torch.from_numpy(lens), torch.from_numpy(pos)
        self.d_out = torch.from_numpy(np.frombuffer(b"".join(outs), np.uint8).copy())
        self.streams = outs
        if timed:
            self.enc_ms.append((time.perf_counter() - t0) * 1e3)


class _StubDecode:
    def __init__(self, w, step):
        self.w, self.step, self.k2_ms, self.ms = w, step, [], []

    def prepare(self):
        pass

    def run(self, timed=True):
        from oracle import orc
        t0 = time.perf_counter()
        self.back = [orc.unrans(s) for s in self.step.streams]
        self.k2_ms.append((time.perf_counter() - t0) * 1e3)
        self.ms.append(self.k2_ms[-1])

    def verify(self):
        w = self.w
        for b, d in zip(w["blocks"], self.back):
            assert d == w["d_in"][b.in_off:b.in_off + b.in_len].numpy().tobytes()
        return "stub round trip"


class StubBackend:
    dist_backend = "gloo"
    seeds = []

    def __init__(self, local):
        self.local = local

    def workload(self, a, seed_shift):
        import bench
        from cramtools_amd import _lib
        StubBackend.seeds.append(seed_shift)
        rng = np.random.default_rng(100 + seed_shift)
        n_cont = max(1, a.reads // 100)
        blocks, parts, off = [], [], 0
        for c in range(n_cont):
            for name, order in bench.SERIES:
                n = 4000 if name == "QS" else 300
                parts.append(rng.integers(0, 7 if name == "QS" else 50, n, dtype=np.uint8))
                blocks.append(_lib.Block(off, 0, n, 0, order, 0))
                pad = (-n) % 16
                parts.append(np.zeros(pad, np.uint8))
                off += n + pad
        parts.append(np.zeros(16, np.uint8))
        rank = int(os.environ.get("RANK", "0"))
        return dict(blocks=blocks, d_in=torch.from_numpy(np.concatenate(parts)), n_cont=n_cont, n_blocks=len(blocks),
                    raw_bytes=int(sum(b.in_len for b in blocks)), delay=0.05 if rank == 1 else 0.0)

    def step(self, w):
        return _StubStep(w)

    def decode_step(self, w, step):
        return _StubDecode(w, step)

    def sync(self):
        pass

    def pcie(self, w):
        return dict(value=1.0, unit="MB/s", note="stub")

    def close(self):
        pass


def test_main_single_process_line(orc, capsys):
    sys.path.insert(0, ROOT)
    import bench
    StubBackend.seeds.clear()
    line = bench.main(["--reads", "400", "--steps", "2", "--warmup", "1"], backend_factory=StubBackend)
    printed = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert printed == json.loads(json.dumps(line))
    assert StubBackend.seeds == [0]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "MB/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    raw = 4 * (4000 + 9 * 300)
    assert abs(line["value"] - raw * 2 / (line["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-6 * line["value"]
    base = line["cpu_baseline"]
    assert base["kind"] == "port" and base["jvm"] in (True, False) and base["nproc"] == os.cpu_count()
    assert base["cores"] >= 1 and base["value"] > 0 and base["decode_value"] > 0
    assert line["parity"].startswith("first")
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["traffic"] is None
    assert line["decode"]["parity"] == "stub round trip" and line["decode"]["value"] > 0
    assert line["pcie_inclusive"]["note"] == "stub"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    import test_bench_ranks as me
    line = bench.main(["--gpus", str(world), "--reads", "300", "--steps", "2", "--warmup", "1"], backend_factory=me.StubBackend)
    q.put((rank, line, list(me.StubBackend.seeds)))


def test_main_two_rank_gloo(orc):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(2):
        r, line, seeds = q.get(timeout=180)
        got[r] = (line, seeds)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    line0, seeds0 = got[0]
    line1, seeds1 = got[1]
    assert line1 is None and seeds0 == [0] and seeds1 == [10]  # one line, from rank 0; the ranks' data differ by seed
    assert line0["n_gpus"] == 2 and "2 rank(s), no collective" in line0["config"]["parallelism"]
    raw = 3 * (4000 + 9 * 300)
    # whole-job aggregate over the MAX of the ranks' times: rank 1 sleeps 50 ms per step, rank 0 does not
    assert line0["ms_per_step"] >= 50.0
    assert abs(line0["value"] - 2 * raw * 2 / (line0["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-6 * line0["value"]
    assert "cpu_baseline" not in line0 and "pcie_inclusive" not in line0
    dec = line0["decode"]
    assert dec["parity"] == "stub round trip"
    assert abs(dec["value"] - 2 * raw / (dec["ms_per_step"] / 1e3) / 1e6) < 1e-6 * dec["value"]
