// Worker threads of a host that keeps the reference's thread model (paralell/TransformerJob.java:29-41): T native
// threads take containers from a shared counter and call the blocking cramgpu_rans_coalesced, one container per call.
// Built as a small shared library that bench.pcie_coalesced drives through ctypes, so that the figure measures the
// library and not the Python interpreter's lock.  Measurement infrastructure, not part of libcramgpu.so.
//   g++ -O2 -shared -fPIC -pthread -I../../include co_driver.cpp -o libco_driver.so
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

#include "cramgpu.h"

typedef int32_t (*coalesced_fn)(cramgpu_coalescer*, int32_t, const uint8_t*, int64_t, uint8_t*, int64_t, const cramgpu_block*,
                                int32_t, int32_t*, int64_t*);

extern "C" double co_driver_threads(coalesced_fn call, cramgpu_coalescer* q, int32_t n_cont, int32_t n_threads, const uint8_t* const* in,
                                    const int64_t* in_bytes, uint8_t* const* out, const int64_t* out_bytes,
                                    const cramgpu_block* const* blocks, int32_t n_blocks, int32_t* const* out_len,
                                    int64_t* const* out_pos, int32_t* first_error) {
    std::atomic<int32_t> next{0}, err{0};
    auto worker = [&] {
        for (;;) {
            const int32_t k = next.fetch_add(1);
            if (k >= n_cont || err.load()) return;
            const int32_t rc = call(q, 0, in[k], in_bytes[k], out[k], out_bytes[k], blocks[k], n_blocks, out_len[k], out_pos[k]);
            if (rc) err.store(rc);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int32_t t = 0; t < n_threads; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *first_error = err.load();
    return s;
}
