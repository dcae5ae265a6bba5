// Probe: what does ds_add_u32 do with an address that is not a multiple of 4 on gfx950?
// RESULT (round 3, MI355X): the process is killed inside the kernel (exit 141, nothing after "launching"): a
// misaligned 32-bit LDS atomic is a memory violation, the low address bits are NOT ignored.  Do not run this next to
// anything that matters.  It ruled out a vote loop of two vector instructions per vote (oslam_vote_body.inc).
// Every lane adds 1 at (word 2*lane) + off bytes, off = 0..3; the words are read back: "low bits dropped" shows as
// word[2*lane] == 1 and a clean neighbour for every off; a true unaligned add shows as 0x100 / 0x10000 / 0x1000000
// or as increments in the neighbouring word.  Also times an aligned against a misaligned stream (replays?).
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_misaligned_atomic.hip -o gpurun_ab/lds_misaligned && gpurun_ab/lds_misaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out, unsigned long long *cyc)
{
    __shared__ uint32_t acc[4 * 256];
    const int lane = threadIdx.x;
    for (int off = 0; off < 4; off++) {
        for (int c = lane; c < 256; c += 64) acc[off * 256 + c] = 0;
        __syncthreads();
        const uint32_t addr = (uint32_t)(uintptr_t)&acc[off * 256 + 2 * lane] + (uint32_t)off;
        const uint32_t one = 1;
        asm volatile("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(one) : "memory");
        __syncthreads();
        for (int c = lane; c < 256; c += 64) out[off * 256 + c] = acc[off * 256 + c];
        __syncthreads();
    }
    for (int off = 0; off < 4; off += 2) {
        uint32_t a[4];
        for (int j = 0; j < 4; j++) a[j] = (uint32_t)(uintptr_t)&acc[(j * 64 + lane)] + (uint32_t)off;
        const uint32_t one = 1;
        __syncthreads();
        const long long t0 = clock64();
        for (int i = 0; i < 4096; i++)
            asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %4\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %4" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(one) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (lane == 0) cyc[off / 2] = (unsigned long long)(t1 - t0);
    }
}
int main()
{
    uint32_t *d_out, h[4 * 256];
    unsigned long long *d_cyc, hc[2];
    setvbuf(stdout, NULL, _IONBF, 0);
    printf("allocating\n");
    if (hipMalloc(&d_out, sizeof h) != hipSuccess || hipMalloc(&d_cyc, sizeof hc) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    printf("launching\n");
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_cyc);
    const hipError_t e = hipDeviceSynchronize();
    printf("synchronised: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hc, d_cyc, sizeof hc, hipMemcpyDeviceToHost);
    for (int off = 0; off < 4; off++) {
        int as_aligned = 0;
        for (int l = 0; l < 64; l++) as_aligned += h[off * 256 + 2 * l] == 1u && h[off * 256 + 2 * l + 1] == 0u;
        printf("offset +%d: %2d of 64 lanes landed as an aligned add of 1; word[0..3] = %08x %08x %08x %08x\n", off, as_aligned,
               h[off * 256], h[off * 256 + 1], h[off * 256 + 2], h[off * 256 + 3]);
    }
    printf("4096 x 4 ds_add_u32 per wave, one wave: aligned %llu cycles, misaligned (+2) %llu cycles\n", hc[0], hc[1]);
    return 0;
}
