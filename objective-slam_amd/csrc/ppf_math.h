/*
 * ppf_math.h -- deterministic float32 math shared by the host C code and the
 * gfx950 kernels of the PPF path.
 *
 * Why this exists: the reference's quantised point-pair feature is hashed
 * byte-for-byte (reference pcl/alignment/src/cuda/kernel.cu:23-30,460-477), so
 * a PPF key is only reproducible if acosf/atan2f return the same bits on the
 * host and on the GPU.  The functions below are float-only operation sequences
 * (add/sub/mul/div/sqrt, each correctly rounded, no fused multiply-add unless
 * written as fmaf) that gcc and hipcc compile to the same IEEE-754 results
 * when built with -ffp-contract=off.  pm_acosf / pm_atanf / pm_atan2f follow
 * the classic fdlibm float algorithms, which is what glibc 2.35 (this image)
 * ships for acosf/atan2f; tests/test_math_exact.py checks them against libm
 * on every float (acosf, atanf) and on >10^9 sampled pairs (atan2f).
 *
 * Build requirement (both compilers): -ffp-contract=off, no -ffast-math,
 * denormals preserved, correctly rounded f32 sqrt/div (hipcc default).
 */
#ifndef OSLAM_PPF_MATH_H
#define OSLAM_PPF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ static inline
#else
#define PM_HD static inline
#endif

#if defined(__cplusplus)
#define PM_BITS_F2U(x) __builtin_bit_cast(uint32_t, (float)(x))
#define PM_BITS_U2F(u) __builtin_bit_cast(float, (uint32_t)(u))
#else
PM_HD uint32_t pm_f2u_(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }
PM_HD float pm_u2f_(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
#define PM_BITS_F2U(x) pm_f2u_(x)
#define PM_BITS_U2F(u) pm_u2f_(u)
#endif

/* float32 constants of the reference: CUDART_PI_F and D_ANGLE0
 * (reference kernel.h:15-16: (2.0f*float(CUDART_PI_F))/float(N_ANGLE)). */
#define PM_PI_F 3.141592654f
#define PM_N_ANGLE 30
#define PM_D_ANGLE ((2.0f * PM_PI_F) / 30.0f)

/* The NaN x86 produces for an invalid operation (0/0, acosf(|x|>1)): negative
 * quiet NaN.  The reference hashes NaN angles byte-wise, so the GPU must use
 * the same bit pattern as the host build it is compared with. */
#define PM_NAN_BITS 0xffc00000u

PM_HD float pm_sqrtf(float x) { return __builtin_sqrtf(x); }
PM_HD float pm_fabsf(float x) { return PM_BITS_U2F(PM_BITS_F2U(x) & 0x7fffffffu); }
PM_HD int pm_isnan(float x) { return (PM_BITS_F2U(x) & 0x7fffffffu) > 0x7f800000u; }

/* The three functions below are written select-style: every lane runs one
 * common operation sequence and picks its own case's operands/results, because
 * a wave whose 64 lanes take different fdlibm branches would otherwise execute
 * every branch (each with its own division).  Per lane the operations and
 * their order are exactly fdlibm's for that lane's case. */

/* acosf: fdlibm e_acosf.c algorithm (rational approximation on three ranges).
 * Returns the canonical NaN for |x| > 1 or NaN input. */
PM_HD float pm_acosf(float x)
{
    const float one = 1.0f;
    const float pi = PM_BITS_U2F(0x40490fdau);
    const float two_pio2_lo = PM_BITS_U2F(0x34222168u);
    const float pio2_hi = PM_BITS_U2F(0x3fc90fdau);
    const float pio2_lo = PM_BITS_U2F(0x33a22168u);
    const float pS0 = PM_BITS_U2F(0x3e2aaaabu), pS1 = PM_BITS_U2F(0xbea6b090u),
                pS2 = PM_BITS_U2F(0x3e4e0aa8u), pS3 = PM_BITS_U2F(0xbd241146u),
                pS4 = PM_BITS_U2F(0x3a4f7f04u), pS5 = PM_BITS_U2F(0x3811ef08u);
    const float qS1 = PM_BITS_U2F(0xc019d139u), qS2 = PM_BITS_U2F(0x4001572du),
                qS3 = PM_BITS_U2F(0xbf303361u), qS4 = PM_BITS_U2F(0x3d9dc62eu);
    const uint32_t hx = PM_BITS_F2U(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const int neg = (int)(hx >> 31);
    const int small = ix < 0x3f000000u;                 /* |x| < 0.5 */
    const float ax = PM_BITS_U2F(ix);
    /* |x| >= 0.5: z = (1 - |x|)/2  (for x < 0 fdlibm writes (one + x)*0.5: same value) */
    const float zb = (one - ax) * 0.5f;
    const float z = small ? x * x : zb;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float s = pm_sqrtf(zb);
    const float df = PM_BITS_U2F(PM_BITS_F2U(s) & 0xfffff000u);
    const float c = (zb - df * df) / (s + df);          /* x > 0.5 only */
    const float res_small = pio2_hi - (x - (pio2_lo - x * r));
    const float res_neg = pi - 2.0f * (s + (r * s - pio2_lo));
    const float res_pos = 2.0f * (df + (r * s + c));
    float res = small ? res_small : (neg ? res_neg : res_pos);
    if (ix <= 0x32800000u) res = pio2_hi + pio2_lo;     /* |x| <= 2^-26 */
    if (ix == 0x3f800000u) res = neg ? pi + two_pio2_lo : 0.0f;
    if (ix > 0x3f800000u) res = PM_BITS_U2F(PM_NAN_BITS);
    return res;
}

/* pm_atanf_pos_, pm_atanf, pm_atan2f and pc_alpha_bin_exact: ppf_math_atan.inc (see there for why it is a
 * separate text) */
#define PM_FN(n) n
#define PM_KF(bits) PM_BITS_U2F(bits)
#define PM_KU(bits) (bits)
/* Exact x - fmodf(x, step) for x >= 0 (or NaN), step > 0:
 * fmodf is exact, so the reference's quant_downf (kernel.cu:90-92) equals
 * RN(k * step) with k = floor(x / step) in exact arithmetic.  k is found with
 * one fmaf-based correction (fmaf keeps the exact sign of x - k*step).
 * *k_out receives k (or -1 for NaN / out-of-range, where the caller must use
 * the generic path). Valid while x / step < 2^21. */
PM_HD float pm_quant_down_pos(float x, float step, float inv_step, int *k_out)
{
    float kf, r;
    if (!(x * inv_step < 2097152.0f)) { /* NaN or huge */
        *k_out = -1;
        return x;
    }
    kf = (float)(int)(x * inv_step);
    r = __builtin_fmaf(-kf, step, x);
    if (r < 0.0f) kf -= 1.0f;
    else if (r >= step) kf += 1.0f;
    *k_out = (int)kf;
    return kf * step;
}

#include "ppf_math_atan.inc"
#undef PM_FN
#undef PM_KF
#undef PM_KU

/* 32-bit FNV-1a over raw bytes read through a SIGNED char, as the reference's
 * hash() does (kernel.cu:23-30; offset basis kernel.h:22): bytes >= 0x80 are
 * sign-extended before the XOR. */
PM_HD uint32_t pm_fnv1a_word(uint32_t h, uint32_t w)
{
    int i;
    for (i = 0; i < 4; i++) {
        h ^= (uint32_t)(int32_t)(int8_t)(w & 0xffu);
        h *= 16777619u;
        w >>= 8;
    }
    return h;
}

#define PM_FNV_BASIS 2166136261u

#endif /* OSLAM_PPF_MATH_H */
