// TEST INFRASTRUCTURE ONLY — runs the kernels of libcramgpu on the CPU (tests/cpu_sim/hip/hip_runtime.h) under
// AddressSanitizer against the CPU oracle.  Usage: sim_main [rans|records|all]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/cramgpu.h"

extern "C" {
int32_t orc_rans_bound(int32_t n);
int32_t orc_rans_compress(const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap);
int32_t orc_rans_uncompress(const uint8_t* in, int32_t n, uint8_t* out, int32_t cap);
}

typedef std::vector<uint8_t> Bytes;

static Bytes orc(const Bytes& d, int order) {
    Bytes o((size_t)orc_rans_bound((int)d.size()));
    int r = orc_rans_compress(d.data(), (int)d.size(), order, o.data(), (int)o.size());
    if (r < 0) { printf("oracle error %d\n", r); exit(2); }
    o.resize((size_t)r);
    return o;
}

static int check_rans(cramgpu_ctx* c, const std::vector<Bytes>& datas, int order) {
    const int n = (int)datas.size();
    std::vector<cramgpu_block> blocks(n);
    int64_t io = 0, oo = 0;
    for (int i = 0; i < n; i++) {
        int len = (int)datas[i].size(), cap = cramgpu_rans_bound(len);
        blocks[i] = cramgpu_block{io, oo, len, cap, order, 0};
        io += (len + 15) & ~15;
        oo += (cap + 15) & ~15;
    }
    Bytes in((size_t)io + 16), out((size_t)oo + 16);
    for (int i = 0; i < n; i++) memcpy(in.data() + blocks[i].in_off, datas[i].data(), datas[i].size());
    std::vector<int32_t> lens(n);
    int rc = cramgpu_rans_encode_batch(c, in.data(), (int64_t)in.size(), out.data(), (int64_t)out.size(), blocks.data(), n, lens.data());
    if (rc) { printf("encode_batch rc=%d %s\n", rc, cramgpu_last_error()); return 1; }
    std::vector<Bytes> streams(n);
    for (int i = 0; i < n; i++) {
        Bytes want = orc(datas[i], order);
        if (lens[i] != (int)want.size() || memcmp(want.data(), out.data() + blocks[i].out_off, want.size())) {
            printf("MISMATCH encode order %d block %d n=%zu got len %d want %zu\n", order, i, datas[i].size(), lens[i], want.size());
            return 1;
        }
        streams[i] = want;
    }
    // decode
    io = oo = 0;
    for (int i = 0; i < n; i++) {
        blocks[i] = cramgpu_block{io, oo, (int)streams[i].size(), (int)datas[i].size(), 0, 0};
        io += (streams[i].size() + 15) & ~15;
        oo += (datas[i].size() + 15) & ~15;
    }
    Bytes din((size_t)io + 16), dout((size_t)oo + 16);
    for (int i = 0; i < n; i++) memcpy(din.data() + blocks[i].in_off, streams[i].data(), streams[i].size());
    rc = cramgpu_rans_decode_batch(c, din.data(), (int64_t)din.size(), dout.data(), (int64_t)dout.size(), blocks.data(), n, lens.data());
    if (rc) { printf("decode_batch rc=%d %s\n", rc, cramgpu_last_error()); return 1; }
    for (int i = 0; i < n; i++) {
        if (lens[i] != (int)datas[i].size() || memcmp(datas[i].data(), dout.data() + blocks[i].out_off, datas[i].size())) {
            printf("MISMATCH decode order %d block %d n=%zu got %d\n", order, i, datas[i].size(), lens[i]);
            return 1;
        }
    }
    return 0;
}

int run_records(cramgpu_ctx* c);

int main(int argc, char** argv) {
    std::string what = argc > 1 ? argv[1] : "all";
    cramgpu_ctx* c = cramgpu_open(0);
    if (!c) { printf("open failed\n"); return 1; }
    int bad = 0;
    if (what == "file") {
        FILE* f = fopen(argv[2], "rb");
        Bytes d(1 << 24);
        d.resize(fread(d.data(), 1, d.size(), f));
        fclose(f);
        std::vector<Bytes> datas{d};
        bad |= check_rans(c, datas, atoi(argv[3]));
        printf("file: %s\n", bad ? "FAILED" : "ok");
    }
    if (what == "rans" || what == "all") {
        std::mt19937 rng(5);
        std::vector<Bytes> datas;
        const bool quick = argc > 2 && std::string(argv[2]) == "quick";
        const std::vector<int> sizes = quick ? std::vector<int>{1, 3, 4, 5, 17, 63, 100, 700}
                                             : std::vector<int>{1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 255, 1000, 4099, 20000};
        const std::vector<int> alph = quick ? std::vector<int>{1, 4, 13, 41, 60, 256} : std::vector<int>{1, 2, 4, 8, 13, 16, 17, 41, 57, 90, 256};
        for (int n : sizes)
            for (int a : alph) {
                Bytes d((size_t)n);
                for (auto& b : d) b = (uint8_t)((rng() % a) * (a <= 8 ? 31 : 1) + (a == 1 ? 65 : 0));
                datas.push_back(d);
            }
        {   // ITF8-like series: a wide alphabet whose contexts have few successors each (ragged order-1 rows in LDS)
            Bytes itf;
            for (int i = 0; i < (quick ? 400 : 6000); i++) {
                int v = 250 + (int)(rng() % 101);
                if (rng() & 1) {
                    itf.push_back((uint8_t)(0x80 | (v >> 8)));
                    itf.push_back((uint8_t)v);
                } else {
                    uint32_t u = (uint32_t)(-v);
                    itf.push_back((uint8_t)(0xF0 | (u >> 28)));
                    itf.push_back((uint8_t)(u >> 20));
                    itf.push_back((uint8_t)(u >> 12));
                    itf.push_back((uint8_t)(u >> 4));
                    itf.push_back((uint8_t)(u & 0x0F));
                }
            }
            datas.push_back(itf);
        }
        Bytes runs(3000);
        for (size_t i = 0; i < runs.size(); i++) runs[i] = (uint8_t)((i / 37) % 5 + 40);
        datas.push_back(runs);
        for (int order = 0; order < 2; order++) bad |= check_rans(c, datas, order);
        printf("rans: %s (%zu blocks x 2 orders)\n", bad ? "FAILED" : "ok", datas.size());
    }
    if (!bad && (what == "records" || what == "all")) bad |= run_records(c);
    cramgpu_close(c);
    return bad;
}
