#!/usr/bin/env python
"""The full per-GPU shard of BASELINE.json configs[3] (600 M reads / 8 GPUs = 75 M x 150 bp reads in 7500 containers over
24 contigs of a 3.1 Gbp / 8 reference share), once through bench.Step (K3 -> F3 -> W -> K1 -> gather) and bench.DecodeStep
(K2 beside G -> R -> N1 / N2) with the round-trip property (decoded arena == raw arena, record columns, features, bases and
scores == the reads); tests/test_configs_gpu.py::test_config4_shard_shape holds the oracle comparison at 2000 containers.

usage: c4_full_shard.py [containers] [contigs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from cramtools_amd import external_compression as xc


def main():
    n_cont = int(sys.argv[1]) if len(sys.argv) > 1 else 7512  # 313 whole containers per contig (7500 = 600 M reads / 8 GPUs / 10 000 rounded up)
    n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    n_reads = n_cont * bench.READS_PER_CONTAINER
    ref_len = 387_500_000 // n_contigs * n_contigs  # this GPU's eighth of 3.1 Gbp, cut into the contigs
    dev = "cuda:0"
    eng = xc.Engine(0)
    t0 = time.perf_counter()
    w = bench.build_workload(dev, n_reads, ref_len, 150, 0, lambda r, ref: bench.gpu_insert_bases(eng, r, ref, n_reads), n_contigs=n_contigs)
    torch.cuda.synchronize()
    print("workload: %d reads, %d containers, %d contigs, %.2f GB of raw rANS series, %.2f GB of bases + scores; built in %.0f s"
          % (n_reads, w["n_cont"], n_contigs, w["raw_bytes"] / 1e9, 2 * w["reads"].bases.numel() / 1e9, time.perf_counter() - t0), flush=True)
    wins = w["reads"].win.cpu().numpy().reshape(-1, 3)
    assert len(set(wins[:, 0].tolist())) == n_contigs
    step = bench.Step(dev, eng, w)
    step.run(False)
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step.run(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("encode step: %.1f ms = %.1f GB/s of payload, rans_encode_kernel %.1f ms" % (dt * 1e3, w["raw_bytes"] / dt / 1e9, step.enc_ms[-1]), flush=True)
    lens = step.d_len.cpu().numpy()
    assert (lens > 0).all(), "a block failed to encode"
    print("streams: %.2f GB (ratio %.2f); positions up to %d" % (lens.sum() / 1e9, w["raw_bytes"] / lens.sum(), int(step.d_pos.max().item())), flush=True)
    dec = bench.DecodeStep(dev, eng, w, step)
    t = time.perf_counter()
    dec.prepare()
    print("decode set-up (zlib level 5 of the GZIP series on the host): %.0f s" % (time.perf_counter() - t), flush=True)
    dec.run(False)
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dec.run(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("decode leg: %.1f ms = %.1f GB/s of payload, rans_decode_kernel %.1f ms" % (dt * 1e3, w["raw_bytes"] / dt / 1e9, dec.k2_ms[-1]), flush=True)
    print("parity:", dec.verify(), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
