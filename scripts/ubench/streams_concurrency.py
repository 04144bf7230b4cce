#!/usr/bin/env python
"""How many kernels from different HIP streams run at once: n streams, one spinning single-workgroup kernel each."""
import time, sys, os
import torch
torch.cuda.init()
cyc = 20_000_000  # about 8-10 ms of spinning at the shader clock
def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
one = torch.cuda.Stream()
run([one]); base = run([one])
print("GPU_MAX_HW_QUEUES=%s: one kernel %.2f ms" % (os.environ.get("GPU_MAX_HW_QUEUES", "unset"), base))
pool = [torch.cuda.Stream() for _ in range(12)]
for n in (2, 3, 4, 6, 8, 12):
    run(pool[:n])
    t = run(pool[:n])
    print("  %2d streams: %.2f ms = %.1f kernels deep" % (n, t, t / base))
