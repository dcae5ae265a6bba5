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

/* atanf of a non-negative finite-or-inf argument, fdlibm s_atanf.c: the four
 * argument reductions are one division num/den with per-range coefficients
 * (num = a*x + b, den = c*x + d; a*x and c*x round exactly as fdlibm's terms). */
PM_HD float pm_atanf_pos_(float ax)
{
    const float aT0 = PM_BITS_U2F(0x3eaaaaabu), aT1 = PM_BITS_U2F(0xbe4ccccdu),
                aT2 = PM_BITS_U2F(0x3e124925u), aT3 = PM_BITS_U2F(0xbde38e38u),
                aT4 = PM_BITS_U2F(0x3dba2e6eu), aT5 = PM_BITS_U2F(0xbd9d8795u),
                aT6 = PM_BITS_U2F(0x3d886b35u), aT7 = PM_BITS_U2F(0xbd6ef16bu),
                aT8 = PM_BITS_U2F(0x3d4bda59u), aT9 = PM_BITS_U2F(0xbd15a221u),
                aT10 = PM_BITS_U2F(0x3c8569d7u);
    const uint32_t ix = PM_BITS_F2U(ax);
    /* range id: -1: <7/16, 0: <11/16, 1: <19/16, 2: <39/16, 3: above */
    const int r0 = ix >= 0x3ee00000u, r1 = ix >= 0x3f300000u, r2 = ix >= 0x3f980000u,
              r3 = ix >= 0x401c0000u;
    const float a = r3 ? 0.0f : ((r0 && !r1) ? 2.0f : 1.0f);
    const float b = r3 ? -1.0f : (r2 ? -1.5f : (r0 ? -1.0f : 0.0f));
    const float c = r3 ? 1.0f : (r2 ? 1.5f : (r0 ? 1.0f : 0.0f));
    const float d = r3 ? 0.0f : (r2 ? 1.0f : (r1 ? 1.0f : (r0 ? 2.0f : 1.0f)));
    const float hi = PM_BITS_U2F(r3 ? 0x3fc90fdau : (r2 ? 0x3f7b985eu : (r1 ? 0x3f490fdau : (r0 ? 0x3eed6338u : 0u))));
    const float lo = PM_BITS_U2F(r3 ? 0x33a22168u : (r2 ? 0x33140fb4u : (r1 ? 0x33222168u : (r0 ? 0x31ac3769u : 0u))));
    const float xr = (a * ax + b) / (c * ax + d);
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    /* with hi = lo = 0 this is x - x*(s1+s2), the |x| < 7/16 form */
    float res = hi - ((xr * (s1 + s2) - lo) - xr);
    if (ix < 0x31000000u) res = ax;                                    /* |x| < 2^-29 */
    if (ix >= 0x4c000000u) res = PM_BITS_U2F(0x3fc90fdau) + PM_BITS_U2F(0x33a22168u);   /* |x| >= 2^25 */
    return res;
}

PM_HD float pm_atanf(float x)
{
    const uint32_t hx = PM_BITS_F2U(x);
    const uint32_t ix = hx & 0x7fffffffu;
    float r;
    if (ix > 0x7f800000u) return x + x;
    r = pm_atanf_pos_(PM_BITS_U2F(ix));
    return PM_BITS_U2F(PM_BITS_F2U(r) | (hx & 0x80000000u));
}

/* atan2f: fdlibm e_atan2f.c algorithm (atanf(|y/x|), quadrant fix-up with a
 * split pi, then the special cases as overriding selects). */
PM_HD float pm_atan2f(float y, float x)
{
    const float tiny = PM_BITS_U2F(0x0da24260u);
    const float pi_o_4 = PM_BITS_U2F(0x3f490fdbu);
    const float pi_o_2 = PM_BITS_U2F(0x3fc90fdbu);
    const float pi = PM_BITS_U2F(0x40490fdbu);
    const float pi_lo = PM_BITS_U2F(0xb3bbbd2eu);
    const uint32_t hx = PM_BITS_F2U(x), hy = PM_BITS_F2U(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const int xneg = (int)(hx >> 31), yneg = (int)(hy >> 31);
    const int k = ((int32_t)iy - (int32_t)ix) >> 23;
    const float z0 = pm_atanf_pos_(pm_fabsf(y / x));
    float z = z0, res;
    if (xneg && k < -60) z = 0.0f;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    {
        const float zl = z - pi_lo;
        const float r01 = PM_BITS_U2F(PM_BITS_F2U(z) ^ ((uint32_t)yneg << 31));
        const float r23 = yneg ? zl - pi : pi - zl;
        res = xneg ? r23 : r01;
    }
    if (iy == 0x7f800000u) res = yneg ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            const float q1 = pi_o_4 + tiny, q3 = 3.0f * pi_o_4 + tiny;
            res = xneg ? (yneg ? -3.0f * pi_o_4 - tiny : q3) : (yneg ? -pi_o_4 - tiny : q1);
        } else {
            res = xneg ? (yneg ? -pi - tiny : pi + tiny) : (yneg ? -0.0f : 0.0f);
        }
    }
    if (ix == 0) res = yneg ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (iy == 0) res = xneg ? (yneg ? -pi - tiny : pi + tiny) : y;
    if (hx == 0x3f800000u) res = PM_BITS_U2F(PM_BITS_F2U(z0) | (hy & 0x80000000u));   /* atanf(y) */
    if (ix > 0x7f800000u || iy > 0x7f800000u) res = x + y;
    return res;
}

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
