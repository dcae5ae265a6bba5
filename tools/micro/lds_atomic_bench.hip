// Microbenchmark: cost of ds_add_u32 (no return) on gfx950 for different address patterns.
// 1 workgroup per CU, 16 waves, each wave issues ITER x 4 atomics; reports cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 4096
__global__ __launch_bounds__(1024) void k(const uint32_t *pat, int npat, unsigned long long *cyc, uint32_t *sink)
{
    __shared__ uint32_t acc[32768 + 128];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int c = tid; c < 32768 + 128; c += 1024) acc[c] = 0;
    __syncthreads();
    // per-lane address tables: 4 patterns per iteration step, rotate through npat sets
    uint32_t a0[4];
    for (int j = 0; j < 4; j++) a0[j] = pat[(j % npat) * 64 + lane];
    const long long t0 = clock64();
    uint32_t rot = 0;
    for (int i = 0; i < ITER; i++) {
        rot = (rot + 0x2080u) & 0x1ffffu;            // move rows (multiple of 128 B) so addresses vary, banks do not
#pragma unroll
        for (int j = 0; j < 4; j++)
            atomicAdd((uint32_t *)((char *)acc + ((a0[j] + rot) & 0x1ffffu)), 1u);
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (acc[tid] == 0xdeadbeefu) sink[0] = 1;
}
int main()
{
    const char *names[] = {"consecutive words (conflict-free)", "bank = lane % 30 (2 lanes share 2 banks per half)", "random bin per lane (30 bins)",
                           "evenly spread angles (lane*30/32)", "all lanes same bank, distinct rows", "all lanes same address",
                           "random bin, random row", "64 distinct banks pattern (lane*4 + (lane>>5)*... )"};
    const int NP = 7;
    uint32_t *d_pat; unsigned long long *d_cyc; uint32_t *d_sink;
    hipMalloc(&d_pat, 4 * 64 * 4); hipMalloc(&d_cyc, 256 * 8); hipMalloc(&d_sink, 4);
    srand(1);
    for (int p = 0; p < NP; p++) {
        std::vector<uint32_t> pat(4 * 64);
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 64; l++) {
                uint32_t row = (uint32_t)(rand() % 1024), bin = 0;
                switch (p) {
                case 0: row = (uint32_t)(2 * j + (l >> 5)); bin = (uint32_t)(l & 31); break;
                case 1: bin = (uint32_t)(l % 30); break;
                case 2: bin = (uint32_t)(rand() % 30); break;
                case 3: bin = (uint32_t)(((l & 31) * 30) / 32); break;
                case 4: bin = 7; break;
                case 5: bin = 7; row = 5; break;
                case 6: bin = (uint32_t)(rand() % 30); break;
                }
                pat[j * 64 + l] = (row * 32 + bin) * 4;
            }
        hipMemcpy(d_pat, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d_pat, 4, d_cyc, d_sink);
            hipDeviceSynchronize();
        }
        unsigned long long h[256]; hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
        s /= 256.0;
        printf("%-55s %8.2f cycles per wave-instruction per CU (16 waves)\n", names[p], s / (16.0 * ITER * 4));
    }
    return 0;
}
