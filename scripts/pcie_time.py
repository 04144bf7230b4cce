#!/usr/bin/env python
"""PCIe-inclusive rate of the asynchronous host-arena boundary (cramgpu_rans_submit / job_wait): containers of the bench
corpus are coded from pinned host arenas, four jobs outstanding, B containers per job."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cramtools_amd import _lib, external_compression as xc

def main():
    n_cont_total, per_job = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reads = n_cont_total * bench.READS_PER_CONTAINER
    dev = "cuda:0"
    eng = xc.Engine(0)
    w = bench.build_workload(dev, reads, reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads))
    ns = len(bench.SERIES)
    d_in = w["d_in"].cpu().numpy()
    L = _lib.lib()
    jobs_desc = []
    for j0 in range(0, n_cont_total, per_job):
        idx = range(j0 * ns, (j0 + per_job) * ns)
        lens = [w["blocks"][i].in_len for i in idx]
        orders = [w["blocks"][i].order for i in idx]
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode(lens, orders)
        h_in, h_out, h_len = L.cramgpu_host_alloc(in_bytes), L.cramgpu_host_alloc(out_bytes), L.cramgpu_host_alloc(4 * len(lens))
        arena = np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), shape=(in_bytes,))
        for k, i in enumerate(idx):
            b = w["blocks"][i]
            arena[blocks[k].in_off:blocks[k].in_off + b.in_len] = d_in[b.in_off:b.in_off + b.in_len]
        jobs_desc.append((blocks, in_bytes, out_bytes, h_in, h_out, h_len, len(lens), sum(lens)))
    def run():
        pending, raw, comp = [], 0, 0
        t0 = time.perf_counter()
        for (blocks, in_bytes, out_bytes, h_in, h_out, h_len, n, rb) in jobs_desc:
            if len(pending) == 4:
                j, hl, nn = pending.pop(0)
                _lib.check(L.cramgpu_job_wait(j))
                comp += int(np.ctypeslib.as_array(C.cast(hl, C.POINTER(C.c_int32)), shape=(nn,)).sum())
            j = L.cramgpu_rans_submit(eng.handle, 0, h_in, in_bytes, h_out, out_bytes, C.addressof(blocks), n, h_len)
            assert j, _lib.last_error()
            pending.append((j, h_len, n))
            raw += rb
        for j, hl, nn in pending:
            _lib.check(L.cramgpu_job_wait(j))
            comp += int(np.ctypeslib.as_array(C.cast(hl, C.POINTER(C.c_int32)), shape=(nn,)).sum())
        return time.perf_counter() - t0, raw, comp
    run()
    for rep in range(3):
        dt, raw, comp = run()
        print("%d containers per job, 4 jobs outstanding: %.1f ms for %.1f MB raw -> %.1f MB  = %.0f MB/s PCIe-inclusive" % (
            per_job, dt * 1e3, raw / 1e6, comp / 1e6, raw / dt / 1e6), flush=True)
    eng.close()

if __name__ == "__main__":
    main()
