#!/usr/bin/env python
"""PCIe-inclusive rate of the asynchronous host-arena boundary (bench.pcie_inclusive) for several job sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cramtools_amd import external_compression as xc

def main():
    n_cont = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    reads = n_cont * bench.READS_PER_CONTAINER
    eng = xc.Engine(0)
    w = bench.build_workload("cuda:0", reads, reads * 64 // 10, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, reads))
    if len(sys.argv) > 2:  # one case only (profiling): containers per job
        r = bench.pcie_inclusive(eng, w, per_job=int(sys.argv[2]), n_cont=n_cont, outstanding=4, reps=1)
        print("%s %d containers per job: %.1f ms = %.0f MB/s" % (r["entry_point"], r["containers_per_job"], r["ms"], r["value"]))
        eng.close()
        return
    for per_job, outstanding, cont in ((4, 4, 128), (4, 16, 256), (16, 4, 256), (16, 16, 512), (64, 4, n_cont), (64, 16, n_cont),
                                       (256, 4, n_cont), (256, -4, n_cont)):
        r = bench.pcie_inclusive(eng, w, per_job=per_job, n_cont=cont, outstanding=abs(outstanding), packed=outstanding > 0)
        print("%-26s %3d containers per job, %2d jobs outstanding, %4d containers: %7.1f ms for %7.1f MB raw -> %6.1f MB = %6.0f MB/s "
              "PCIe-inclusive" % (r["entry_point"], per_job, abs(outstanding), r["containers"], r["ms"], r["raw_MB"],
                                  r["compressed_MB"], r["value"]), flush=True)
    eng.close()

if __name__ == "__main__":
    main()
