// Microbenchmark: issue cost of the integer vector instructions the vote loop uses, gfx950.
// 16 waves per CU (4 per SIMD), each runs a dependent-free stream of N copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 2000
#define BODY(INSTR)                                                                                  \
    for (int i = 0; i < ITER; i++) {                                                                 \
        _Pragma("unroll") for (int k = 0; k < REP / 8; k++) {                                        \
            asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)     \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),   \
                           "+v"(r[6]), "+v"(r[7])                                                    \
                         : "v"(a), "v"(b));                                                          \
        }                                                                                            \
    }
#define I_MULHI24(n) "v_mul_hi_u32_u24 %" #n ", 0x1e00, %8\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", 0x1e00, %8\n"
#define I_MULHI32(n) "v_mul_hi_u32 %" #n ", %8, %9\n"
#define I_MULLO32(n) "v_mul_lo_u32 %" #n ", %8, %9\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %8, %9\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 2, %9\n"
#define I_MIN3(n) "v_min3_u32 %" #n ", %8, %9, %" #n "\n"
#define I_MADU24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n"
#define I_MULF32(n) "v_mul_f32 %" #n ", %8, %9\n"
#define I_CVT(n) "v_cvt_f32_u32 %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, 0\n"
#define I_ADD(n) "v_add_u32 %" #n ", %8, %9\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 2, %9\n"
#define I_AND(n) "v_and_b32 %" #n ", %8, %9\n"
#define I_MIN(n) "v_min_u32 %" #n ", %8, %9\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_CMP(n) "v_cmp_gt_u32 vcc, %8, %" #n "\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %8, 10, 1\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n"
#define I_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %8, %9, 2\n"
#define I_MAD64A(n) "v_mad_u64_u32 %" #n ", vcc, %8, 30, %" #n "\n"
template <int W>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, uint32_t *sink)
{
    uint32_t r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    uint32_t a = threadIdx.x * 2654435761u, b = threadIdx.x + 17;
    __syncthreads();
    const long long t0 = clock64();
    if (W == 0) { BODY(I_MULHI24) }
    if (W == 1) { BODY(I_MUL24) }
    if (W == 2) { BODY(I_MULHI32) }
    if (W == 3) { BODY(I_MULLO32) }
    if (W == 4) { BODY(I_SUB) }
    if (W == 5) { BODY(I_LSHLADD) }
    if (W == 6) { BODY(I_MIN3) }
    if (W == 7) { BODY(I_MADU24) }
    if (W == 8) { BODY(I_MULF32) }
    if (W == 9) { BODY(I_CVT) }
    if (W == 10) { BODY(I_FMA) }
    if (W == 12) { BODY(I_ADD) }
    if (W == 13) { BODY(I_LSHL) }
    if (W == 14) { BODY(I_AND) }
    if (W == 15) { BODY(I_MIN) }
    if (W == 16) { for (int i = 0; i < ITER; i++) { _Pragma("unroll") for (int k = 0; k < REP / 8; k++) asm volatile(I_CNDMASK(0) I_CNDMASK(1) I_CNDMASK(2) I_CNDMASK(3) I_CNDMASK(4) I_CNDMASK(5) I_CNDMASK(6) I_CNDMASK(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "vcc"); } }
    if (W == 17) { for (int i = 0; i < ITER; i++) { _Pragma("unroll") for (int k = 0; k < REP / 8; k++) asm volatile(I_CMP(0) I_CMP(1) I_CMP(2) I_CMP(3) I_CMP(4) I_CMP(5) I_CMP(6) I_CMP(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "vcc"); } }
    if (W == 18) { BODY(I_BFE) }
    if (W == 19) { BODY(I_ADD3) }
    if (W == 20) { for (int i = 0; i < ITER; i++) { _Pragma("unroll") for (int k = 0; k < REP / 8; k++) asm volatile(I_READLANE(0) I_READLANE(1) I_READLANE(2) I_READLANE(3) I_READLANE(4) I_READLANE(5) I_READLANE(6) I_READLANE(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "s20"); } }
    if (W == 21) { BODY(I_MOV) }
    if (W == 22) { BODY(I_ADDLSHL) }
    if (W == 23) {
        unsigned long long q[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int i = 0; i < ITER; i++) {
#pragma unroll
            for (int k = 0; k < REP / 8; k++) {
                asm volatile(I_MAD64A(0) I_MAD64A(1) I_MAD64A(2) I_MAD64A(3) I_MAD64A(4) I_MAD64A(5) I_MAD64A(6) I_MAD64A(7)
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                             : "v"(a), "v"(b) : "vcc");
            }
        }
        r[0] += (uint32_t)(q[0] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + q[7]);
    }
    if (W == 11) {
        unsigned long long q[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int i = 0; i < ITER; i++) {
#pragma unroll
            for (int k = 0; k < REP / 8; k++) {
                asm volatile(I_MAD64(0) I_MAD64(1) I_MAD64(2) I_MAD64(3) I_MAD64(4) I_MAD64(5) I_MAD64(6) I_MAD64(7)
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                             : "v"(a), "v"(b) : "vcc");
            }
        }
        r[0] += (uint32_t)(q[0] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + q[7]);
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] == 0x1234567u) sink[0] = 1;
}
template <int W>
void run(const char *name, unsigned long long *d_cyc, uint32_t *d_sink)
{
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<W>, dim3(256), dim3(1024), 0, 0, d_cyc, d_sink); (void)hipDeviceSynchronize(); }
    unsigned long long h[256]; (void)hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
    s /= 256.0;
    // per SIMD: 4 waves x ITER x REP instructions
    printf("%-30s %6.2f cycles per wave-instruction per SIMD (4 waves per SIMD)\n", name, s / (4.0 * ITER * REP));
}
int main()
{
    unsigned long long *d_cyc; uint32_t *d_sink;
    (void)hipMalloc(&d_cyc, 256 * 8); (void)hipMalloc(&d_sink, 4);
    run<0>("v_mul_hi_u32_u24", d_cyc, d_sink);
    run<1>("v_mul_u32_u24", d_cyc, d_sink);
    run<2>("v_mul_hi_u32", d_cyc, d_sink);
    run<3>("v_mul_lo_u32", d_cyc, d_sink);
    run<4>("v_sub_u32", d_cyc, d_sink);
    run<5>("v_lshl_add_u32", d_cyc, d_sink);
    run<6>("v_min3_u32", d_cyc, d_sink);
    run<7>("v_mad_u32_u24", d_cyc, d_sink);
    run<8>("v_mul_f32", d_cyc, d_sink);
    run<9>("v_cvt_f32_u32", d_cyc, d_sink);
    run<10>("v_fma_f32", d_cyc, d_sink);
    run<11>("v_mad_u64_u32", d_cyc, d_sink);
    run<23>("v_mad_u64_u32 + vgpr addend", d_cyc, d_sink);
    run<12>("v_add_u32", d_cyc, d_sink);
    run<13>("v_lshlrev_b32", d_cyc, d_sink);
    run<14>("v_and_b32", d_cyc, d_sink);
    run<15>("v_min_u32", d_cyc, d_sink);
    run<16>("v_cndmask_b32", d_cyc, d_sink);
    run<17>("v_cmp_gt_u32", d_cyc, d_sink);
    run<18>("v_bfe_u32", d_cyc, d_sink);
    run<19>("v_add3_u32", d_cyc, d_sink);
    run<20>("v_readlane_b32", d_cyc, d_sink);
    run<21>("v_mov_b32", d_cyc, d_sink);
    run<22>("v_add_lshl_u32", d_cyc, d_sink);
    return 0;
}
