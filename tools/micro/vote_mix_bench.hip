// Microbenchmark: one vote iteration of k_vote (readlane + 4 x (v_sub, v_mad_u64_u32, v_lshl_add) + v_min, v_min3,
// v_cmp + 4 ds_add_u32) on gfx950, 16 waves per CU, with its parts switched on and off:
//   W = 0  the 16 vector instructions alone         W = 1  the 4 LDS atomics alone (conflict-free addresses)
//   W = 2  both, as the kernel issues them          W = 3  both, addresses from the arithmetic (random banks)
//   W = 4  fast-mode mix (13 vector instructions) + atomics
//   W = 5  as 2 with two workgroups of 8 waves per CU (grid 512 x 512 threads)
// Prints SIMD cycles per iteration per wave (wall cycles x 4 waves per SIMD would be the naive bound).
// hipcc --offload-arch=gfx950 -O3 tools/micro/vote_mix_bench.hip -o /tmp/vote_mix && /tmp/vote_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
template <int W>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, uint32_t *sink)
{
    constexpr int ROWS = W == 5 ? 512 : 1024;
    __shared__ uint32_t acc[31 * ROWS + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int c = tid; c < 31 * ROWS + 64; c += blockDim.x) acc[c] = 0;
    __syncthreads();
    uint32_t wa[4], rowb[4], inc[4], fixed[4];
    for (int j = 0; j < 4; j++) {
        wa[j] = (uint32_t)(tid * 2654435761u + j * 40503u);
        rowb[j] = (uint32_t)(uintptr_t)acc + ((wa[j] & (uint32_t)(ROWS - 1)) * 124u);
        inc[j] = (wa[j] >> 10) & 1u ? 0x10000u : 1u;
        fixed[j] = (uint32_t)(uintptr_t)acc + 4u * (uint32_t)((j * 64 + lane) + (tid >> 6) * 256);   // consecutive words
    }
    uint32_t csmv = (uint32_t)tid * 747796405u;
    uint32_t hit = 0;
    unsigned long long trig = 0;
    const long long t0 = clock64();
    for (int i = 0; i < ITER; i++) {
        uint32_t addr[4], pos[4];
        if (W != 1) {
            const uint32_t csm = (uint32_t)__builtin_amdgcn_readlane((int)csmv, i & 63) + (uint32_t)i * 0x9e3779b9u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned long long p = (unsigned long long)(csm - wa[j]) * 30ull;
                uint32_t bin = (uint32_t)(p >> 32);
                pos[j] = (uint32_t)p;
                asm("" : "+v"(bin));
                addr[j] = rowb[j] + (bin << 2);
            }
            if (W != 4) {
                const uint32_t lo3 = min(min(pos[0], pos[1]), pos[2]);
                const unsigned long long near = __ballot(min(lo3, pos[3]) < 16u);
                if (__builtin_expect(near != 0ull, 0)) trig += near;
            }
        }
        if (W == 0) {
            asm volatile("" ::"v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]));
        } else if (W == 1 || W == 2 || W == 5) {
            asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %5\n\tds_add_u32 %2, %6\n\tds_add_u32 %3, %7"
                         :: "v"(fixed[0]), "v"(fixed[1]), "v"(fixed[2]), "v"(fixed[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3]) : "memory");
            if (W != 1) asm volatile("" ::"v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]));
        } else {
            asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %5\n\tds_add_u32 %2, %6\n\tds_add_u32 %3, %7"
                         :: "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3]) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (acc[tid] == 0xdeadbeefu || trig == 0x1234567ull) sink[0] = hit + 1;
}
template <int W>
static void run(const char *name, int grid, int block, unsigned long long *d_cyc, uint32_t *d_sink)
{
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<W>, dim3(grid), dim3(block), 0, 0, d_cyc, d_sink);
        hipDeviceSynchronize();
    }
    static unsigned long long h[1024];
    hipMemcpy(h, d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; i++) s += (double)h[i];
    s /= grid;
    printf("%-70s %8.1f wall cycles per iteration per wave (16 waves per CU)\n", name, s / ITER);
}
int main()
{
    unsigned long long *d_cyc;
    uint32_t *d_sink;
    hipMalloc(&d_cyc, 1024 * 8);
    hipMalloc(&d_sink, 4);
    run<0>("16 vector instructions (exact-mode iteration), no atomics", 256, 1024, d_cyc, d_sink);
    run<1>("4 ds_add_u32 alone, conflict-free", 256, 1024, d_cyc, d_sink);
    run<2>("both, conflict-free atomics", 256, 1024, d_cyc, d_sink);
    run<3>("both, atomics at the computed (random-bank) addresses", 256, 1024, d_cyc, d_sink);
    run<4>("fast-mode mix (13 vector instructions) + atomics at computed addresses", 256, 1024, d_cyc, d_sink);
    run<5>("both, conflict-free, two workgroups of 8 waves per CU", 512, 512, d_cyc, d_sink);
    return 0;
}
