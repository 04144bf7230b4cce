#!/usr/bin/env python
"""Re-runs the random-container parity test of tests/test_records_gpu.py (K3 / matrices / K4 / K5 / merge against the
oracle, short and long reads) with many seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_records_gpu as t
from cramtools_amd import external_compression as xc, records
from oracle import orc

def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    orc.build()
    eng = xc.Engine(0)
    real = np.random.default_rng
    try:
        for k in range(n_seeds):
            np.random.default_rng = lambda s=None, k=k: real(1000 * (k + 1) + (s or 0))
            for long_reads in (False, True):
                try:
                    t.test_random_containers_vs_oracle((records, eng), orc, long_reads)
                except Exception as e:
                    # a container on which ReferenceTracks itself throws (the oracle raises the same, see
                    # tests/test_host_entry_points.py): everything before the lossy model has been compared by then
                    if "ReferenceTracks would throw" not in str(e):
                        raise
                    print("seed %d long=%s: %s (reference exception, skipped the lossy part)" % (k, long_reads, e), flush=True)
            t.test_multi_container_batch((records, eng), orc)
            t.test_restore_quals_many_reads((records, eng), orc)
            print("seed %d ok" % k, flush=True)
    finally:
        np.random.default_rng = real
        eng.close()
    print("records fuzz ok: %d seeds x (short, long, multi-container, quality restoration)" % n_seeds)

if __name__ == "__main__":
    main()
