"""cramtools_amd: host-side mirror of the reference's interface over libcramgpu.so (include/cramgpu.h)."""
import os

# Hardware queues per process: HIP maps streams onto GPU_MAX_HW_QUEUES queues (default 4) round robin, and launches
# of streams that share a queue run one after the other.  The asynchronous job slots and the coalescing queue of the
# library live on overlapping launches of several streams (scripts/ubench/streams_concurrency.py: 3 kernels at once
# with 4 queues, 7 with 8), so the package asks for 8 unless the process was told otherwise.  Read by the HIP runtime
# when it initialises, i.e. at the first HIP call of the process; libcramgpu.so does the same for JNI hosts when it is
# loaded (csrc/api.hip).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
