// Microbenchmark + functional check: v_sub_u32_dpp with row_newbcast:N on gfx950 (lane N of every row of 16 lanes
// supplies src0) -- what the vote loop could use instead of v_readlane + v_sub.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_func(uint32_t *o, const uint32_t *a, const uint32_t *b)
{
    uint32_t x = a[threadIdx.x], y = b[threadIdx.x], r5, r15, r0;
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(r5) : "v"(x), "v"(y));
    asm volatile("v_sub_u32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(r15) : "v"(x), "v"(y));
    asm volatile("v_sub_u32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(r0) : "v"(x), "v"(y));
    o[threadIdx.x] = r5;
    o[64 + threadIdx.x] = r15;
    o[128 + threadIdx.x] = r0;
}
#define ITER 2000
template <int W>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long *cyc, uint32_t *sink)
{
    uint32_t r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    uint32_t a = threadIdx.x * 2654435761u, b = threadIdx.x + 17;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (W == 0)
                asm volatile("v_sub_u32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_u32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_u32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_u32_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
            else
                asm volatile("v_sub_u32 %0, %8, %9\n v_sub_u32 %1, %8, %9\n v_sub_u32 %2, %8, %9\n v_sub_u32 %3, %8, %9\n"
                             "v_sub_u32 %4, %8, %9\n v_sub_u32 %5, %8, %9\n v_sub_u32 %6, %8, %9\n v_sub_u32 %7, %8, %9\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] == 0x1234567u) sink[0] = 1;
}
template <int W>
void run(const char *name, unsigned long long *d_cyc, uint32_t *d_sink)
{
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_rate<W>, dim3(256), dim3(1024), 0, 0, d_cyc, d_sink); (void)hipDeviceSynchronize(); }
    unsigned long long h[256]; (void)hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
    printf("%-28s %6.2f cycles per wave-instruction per SIMD (4 waves per SIMD)\n", name, s / 256.0 / (4.0 * ITER * 64));
}
int main()
{
    uint32_t ha[64], hb[64], ho[192], *da, *db, *dout;
    for (int i = 0; i < 64; i++) { ha[i] = 1000u * i + 7u; hb[i] = i; }
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dout, 768);
    (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_func, dim3(1), dim3(64), 0, 0, dout, da, db);
    (void)hipMemcpy(ho, dout, 768, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        bad += ho[i] != ha[(i & ~15) + 5] - hb[i];
        bad += ho[64 + i] != ha[(i & ~15) + 15] - hb[i];
        bad += ho[128 + i] != ha[(i & ~15) + 0] - hb[i];
    }
    printf("row_newbcast semantics (lane N of the lane's own row of 16): %s\n", bad ? "DIFFERENT" : "as expected");
    unsigned long long *d_cyc; uint32_t *d_sink;
    (void)hipMalloc(&d_cyc, 256 * 8); (void)hipMalloc(&d_sink, 4);
    run<0>("v_sub_u32_dpp row_newbcast", d_cyc, d_sink);
    run<1>("v_sub_u32", d_cyc, d_sink);
    return bad != 0;
}
