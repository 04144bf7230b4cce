#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    __shared__ unsigned big[160*256-64];
    big[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    for (int i = 0; i < 100000; i++) asm volatile("s_sleep 1");
    if (big[threadIdx.x ^ 1] == 77) out[0] = 0;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 16 * 4); hipMemset(d, 0xFF, 4096*16*4);
    for (int nw : {9, 12}) {
        hipLaunchKernelGGL(k, dim3(512), dim3(64 * nw), 0, 0, d);
        hipDeviceSynchronize();
        unsigned h[512 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int hist[16][4] = {};
        for (int b = 0; b < 512; b++) for (int w = 0; w < nw; w++) hist[w][(h[b*16+w] >> 4) & 3]++;
        printf("waves %d: SIMD histogram per wave index over 512 workgroups\n", nw);
        for (int w = 0; w < nw; w++) printf("  wave %2d: %4d %4d %4d %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        for (int b = 0; b < 4; b++) { printf("  wg %d:", b); for (int w = 0; w < nw; w++) printf(" %u", (h[b*16+w] >> 4) & 3); printf("  cu %u se %u\n", (h[b*16]>>8)&15, (h[b*16]>>13)&7); }
    }
}
