#!/usr/bin/env python
"""CRAMGPU_PROF=1 run of a few encode launches (1 quality block, 1000 quality blocks): in-kernel cycle counters."""
import os, sys
os.environ["CRAMGPU_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cramtools_amd import _lib, external_compression as xc, synth

def main():
    eng = xc.Engine(0)
    import bench
    q = synth.markov_quals(1500000, 2)
    # the bench's own quality stream (first container), to compare with the plain markov stream
    from cramtools_amd import synth_torch
    ref = synth_torch.reference(640000, device="cuda:0")
    rs = synth_torch.make_reads(10000, ref, read_len=150, seed=1, qual_seed=2, err_seed=3)
    qb = rs.quals[:1500000].cpu().numpy()
    print("alphabet markov %d, bench %d" % (len(set(q.tolist())), len(set(qb.tolist()))))
    for q, nb in ((q, 1), (qb, 1), (qb, 1000)):
        n = q.size
        lens, orders = [n] * nb, [1] * nb
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode(lens, orders)
        d_in = torch.zeros(in_bytes, dtype=torch.uint8, device="cuda:0")
        tq = torch.as_tensor(q).to("cuda:0")
        for i in range(nb):
            d_in[blocks[i].in_off:blocks[i].in_off + n] = tq
        d_out = torch.empty(out_bytes, dtype=torch.uint8, device="cuda:0")
        d_len = torch.zeros(nb, dtype=torch.int32, device="cuda:0")
        d_pos = torch.zeros(nb, dtype=torch.int64, device="cuda:0")
        for rep in range(2):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            eng.encode_dev(d_in, d_out, blocks, nb, d_len, d_pos)
            t1.record(); torch.cuda.synchronize()
            print("blocks %d: %.3f ms" % (nb, t0.elapsed_time(t1)), flush=True)
    eng.close()

main()
