#!/usr/bin/env python
"""PCIe-inclusive rate of the coalescing queue at the reference's own job size: ONE container per submission
(bench.pcie_coalesced), configs[1] corpus.

usage: coalesce_time.py [containers] [max_batches]   (CO_THREADS / CO_TICKETS: comma-separated counts)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cramtools_amd import external_compression as xc


def main():
    n_cont = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    max_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    reads = n_cont * bench.READS_PER_CONTAINER
    eng = xc.Engine(0)
    w = bench.build_workload("cuda:0", reads, reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads))
    tc = tuple(int(x) for x in os.environ.get("CO_THREADS", "1,16,64,128,256").split(",") if x)
    wc = tuple(int(x) for x in os.environ.get("CO_TICKETS", "16,64,128,256").split(",") if x)
    print("# %d containers of configs[1], 1 container per submission, batch slots %s, GPU_MAX_HW_QUEUES=%s" %
          (n_cont, max_batches or "default", os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
    for r in bench.pcie_coalesced(eng, w, n_cont, tc, wc, max_batches):
        print("%-22s %3d %s outstanding, %4d containers: %7.1f ms for %7.1f MB raw -> %6.1f MB = %6.0f MB/s PCIe-inclusive, "
              "%.1f containers per launch" % (r["entry_point"], r["outstanding"], r["driver"], r["containers"], r["ms"], r["raw_MB"],
                                              r["compressed_MB"], r["value"], r["containers_per_launch"]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
