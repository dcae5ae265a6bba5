/*
 * ppf_core.h -- the point-pair feature key and the alpha bin, written once for
 * the host C code and the gfx950 kernels (product code; the CPU oracle has its
 * own, independent restatement under oracle/).
 *
 * Follows the reference's arithmetic operation by operation
 * (pcl/alignment/src/cuda/kernel.cu): compute_ppf :109-122, disc_feature
 * :94-100, ppf_hash_kernel :460-477, hash :23-30, and the alpha part of
 * trans_model_scene :338-342.
 */
#ifndef OSLAM_PPF_CORE_H
#define OSLAM_PPF_CORE_H

#include "ppf_acos_table.h"
#include "ppf_alpha_table.h"
#include "ppf_math.h"

#include <math.h>
/* fmodf is exact by definition; libm on the host, the HIP device library on the GPU */
#define PC_FMODF(x, y) fmodf((x), (y))

/* x - fmodf(x, step) (kernel.cu:90-92) for x >= 0 or NaN; bits with the host's
 * NaN pattern so that hashing the bytes matches an x86 build. */
PM_HD uint32_t pc_quant_bits(float x, float step, float inv_step)
{
    int k;
    float q = pm_quant_down_pos(x, step, inv_step, &k);
    if (k < 0) q = x - PC_FMODF(x, step);      /* NaN, inf or an absurd quotient */
    if (pm_isnan(q)) return PM_NAN_BITS;
    return PM_BITS_F2U(q);
}

/* Distance part of a pair's feature: |d| and its bin k1 = floor(|d| / d_dist) (or -1 when
 * the quotient is NaN or >= 2^21 and the generic quantisation has to run). */
PM_HD int pc_pair_dist_bin(float dx, float dy, float dz, float d_dist, float inv_d_dist)
{
    int k;
    (void)pm_quant_down_pos(pm_sqrtf(dx * dx + dy * dy + dz * dz), d_dist, inv_d_dist, &k);
    return k;
}

/* Key of the ordered pair (p1,n1) -> (p2,n2); 0 when the distance is not a
 * finite number (the reference maps NaN in .x to key 0, kernel.cu:467-469).
 * n1n = norm(n1) is passed in because the caller keeps it per point. */
PM_HD uint32_t pc_pair_key(float p1x, float p1y, float p1z, float n1x, float n1y, float n1z,
                           float n1n, float p2x, float p2y, float p2z, float n2x, float n2y,
                           float n2z, float n2n, float d_dist, float inv_d_dist)
{
    const float D = PM_D_ANGLE;
    const float invD = 1.0f / PM_D_ANGLE;
    float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    float nd = pm_sqrtf(dx * dx + dy * dy + dz * dz);
    float a2 = pm_acosf((n1x * dx + n1y * dy + n1z * dz) / (n1n * nd));
    float a3 = pm_acosf((n2x * dx + n2y * dy + n2z * dz) / (n2n * nd));
    float a4 = pm_acosf((n1x * n2x + n1y * n2y + n1z * n2z) / (n1n * n2n));
    int k;
    float q1 = pm_quant_down_pos(nd, d_dist, inv_d_dist, &k);
    uint32_t h;
    if (k < 0) {
        q1 = nd - PC_FMODF(nd, d_dist);
        if (pm_isnan(q1)) return 0u;
    }
    h = pm_fnv1a_word(PM_FNV_BASIS, PM_BITS_F2U(q1));
    h = pm_fnv1a_word(h, pc_quant_bits(a2, D, invD));
    h = pm_fnv1a_word(h, pc_quant_bits(a3, D, invD));
    h = pm_fnv1a_word(h, pc_quant_bits(a4, D, invD));
    return h;
}

/* floor(acosf(c) / D) of pc_pair_key without evaluating acosf: the bin is a step function of the float c whose
 * steps were tabulated on every float (ppf_acos_table.h; `lut` = PC_ACOS_LUT or a copy of it).  0..15, or 16
 * where acosf gives NaN (the reference then hashes the NaN pattern). */
PM_HD uint32_t pc_acos_bin(float c, const uint32_t *lut)
{
    int cell;
    if (!(pm_fabsf(c) <= 1.0f)) return 16u;
    cell = (int)(c * 64.0f + 64.0f);
    cell = cell > PC_ACOS_CELLS - 1 ? PC_ACOS_CELLS - 1 : cell;
    return lut[2 * cell + 1] + (c <= PM_BITS_U2F(lut[2 * cell]) ? 1u : 0u);
}

/* The quantised feature of the ordered pair as bins instead of a hash: returns the distance bin k1 and
 * *combo = j2 + 17*j3 + 289*j4 with pc_pair_key(...) == pc_key_of_bins(k1, *combo, d_dist); -1 where
 * pc_pair_key takes its generic path (distance NaN, infinite or >= 2^21 bins) and the caller has to as well.
 * Same operations on the same operands as pc_pair_key up to each acosf argument. */
PM_HD int pc_pair_bins(float p1x, float p1y, float p1z, float n1x, float n1y, float n1z, float n1n, float p2x,
                       float p2y, float p2z, float n2x, float n2y, float n2z, float n2n, float d_dist,
                       float inv_d_dist, const uint32_t *lut, uint32_t *combo)
{
    float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    float nd = pm_sqrtf(dx * dx + dy * dy + dz * dz);
    const uint32_t j2 = pc_acos_bin((n1x * dx + n1y * dy + n1z * dz) / (n1n * nd), lut);
    const uint32_t j3 = pc_acos_bin((n2x * dx + n2y * dy + n2z * dz) / (n2n * nd), lut);
    const uint32_t j4 = pc_acos_bin((n1x * n2x + n1y * n2y + n1z * n2z) / (n1n * n2n), lut);
    int k;
    (void)pm_quant_down_pos(nd, d_dist, inv_d_dist, &k);
    *combo = j2 + 17u * j3 + 289u * j4;
    return k;
}

/* Every key a pair in distance bin k1 can have: the quantised distance is (float)k1 * d_dist
 * and each quantised angle is one of (float)j * D, j = 0..15, or the NaN pattern -- 17^3
 * combinations.  `combo` in [0, 4913). */
#define PC_ANGLE_VALUES 17
#define PC_ANGLE_COMBOS (17 * 17 * 17)
PM_HD uint32_t pc_key_of_bins(uint32_t k1, uint32_t combo, float d_dist)
{
    const uint32_t j2 = combo % 17u, j3 = (combo / 17u) % 17u, j4 = combo / 289u;
    uint32_t h = pm_fnv1a_word(PM_FNV_BASIS, PM_BITS_F2U((float)k1 * d_dist));
    h = pm_fnv1a_word(h, j2 < 16u ? PM_BITS_F2U((float)j2 * PM_D_ANGLE) : PM_NAN_BITS);
    h = pm_fnv1a_word(h, j3 < 16u ? PM_BITS_F2U((float)j3 * PM_D_ANGLE) : PM_NAN_BITS);
    h = pm_fnv1a_word(h, j4 < 16u ? PM_BITS_F2U((float)j4 * PM_D_ANGLE) : PM_NAN_BITS);
    return h;
}

/* norm(n) as the reference computes it (kernel.cu:51-61) */
PM_HD float pc_norm3(float x, float y, float z) { return pm_sqrtf(x * x + y * y + z * z); }

/* y and z of T_g * (p, 1) with T_g rows given as 4 floats each
 * (mat4f_vmul / dot(float4,float4), kernel.cu:55-57,234-242) */
PM_HD float pc_row_dot(const float *row, float px, float py, float pz)
{
    return row[0] * px + row[1] * py + row[2] * pz + row[3] * 1.0f;
}

/* pc_alpha_bin_exact (alpha bin of kernel.cu:338-342 by the full formula): ppf_math_atan.inc */

/* The same bin without evaluating atan2f: in atan2f's main path the result is a
 * function of q = fl(|cross/dot|) and the two signs only, and the bin is a monotone
 * step function of q whose steps (ppf_alpha_table.h) were found by evaluating the
 * exact formula on every float q.  One division, four table reads.  `tbl` is the
 * table PC_ALPHA_THR (32 words; the kernels keep a copy in LDS, one word per bank).
 * Inputs outside the main path (zero, infinite, NaN or 2^60 apart) take the full
 * formula. */
#define PC_ALPHA_OUTSIDE 0xffffffffu      /* pc_alpha_bin_table_main: the input is outside the main path */
PM_HD unsigned pc_alpha_bin_table_main(float uy, float uz, float vy, float vz, const uint32_t *tbl)
{
    const float cx = uy * vz - uz * vy;
    const float dt = 0.0f * 0.0f + uy * vy + uz * vz;
    const uint32_t hy = PM_BITS_F2U(cx), hx = PM_BITS_F2U(dt);
    const uint32_t iy = hy & 0x7fffffffu, ix = hx & 0x7fffffffu;
    const int k = ((int32_t)iy - (int32_t)ix) >> 23;
    /* iy, ix in [1, 0x7f7fffff] and |k| <= 60 */
    if ((iy - 1u) < 0x7f7fffffu && (ix - 1u) < 0x7f7fffffu && (unsigned)(k + 60) <= 120u) {
        const uint32_t qb = PM_BITS_F2U(PM_BITS_U2F(iy) / PM_BITS_U2F(ix));
        const unsigned m = (hy >> 31) | ((hx >> 31) << 1);
        const uint32_t *row = tbl + 8 * m;
        unsigned pos = (row[4] <= qb) ? 4u : 0u;
        const unsigned tiny = row[0] <= qb;
        pos += (row[pos + 2] <= qb) ? 2u : 0u;
        pos += (row[pos + 1] <= qb) ? 1u : 0u;
        pos += tiny;
        /* PC_ALPHA_BASE = {15, 15, 30, 0}; + for quadrants 0 and 3, - for 1 and 2 */
        return (m == 0u) ? 15u + pos : (m == 1u) ? 15u - pos : (m == 2u) ? 30u - pos : pos;
    }
    return PC_ALPHA_OUTSIDE;
}
PM_HD unsigned pc_alpha_bin_table(float uy, float uz, float vy, float vz, const uint32_t *tbl)
{
    const unsigned b = pc_alpha_bin_table_main(uy, uz, vy, vz, tbl);
    return b != PC_ALPHA_OUTSIDE ? b : pc_alpha_bin_exact(uy, uz, vy, vz);
}

/* ---- quantised-angle voting -------------------------------------------------
 * In exact arithmetic alpha = theta_v - theta_u with theta = atan2(z, y) of the
 * transformed second points (Drost).  A model pair stores theta_u, a scene pair
 * theta_v, both as theta + pi in units of 2^-22 turn (7.2e-6 bin), so their
 * difference wraps by itself and
 *     t24 = 4 * ((theta_v - theta_u + half a turn) mod one turn)        (24 bits)
 *     t24 * 7680 = bin * 2^32 + position inside the bin * 2^32          (7680 = 30 * 2^8)
 * i.e. the bin is v_mul_hi_u32_u24(t24, 7680) and the position v_mul_u32_u24(t24, 7680).
 * The float rounding of the reference's own sequence (cross/dot products, atan2f,
 * + pi, quantisation) moves alpha by < 2e-5 bin against this value (bound in
 * DESIGN.md), so whenever the position is further than the margin (1.14e-4 bin)
 * from a bin edge the bin is the reference's; otherwise the vote is re-evaluated
 * with pc_alpha_bin_table.  The margin is added to t24 first, so that one unsigned
 * compare of the position finds both sides of an edge; the bin read from the
 * shifted value is only used when the vote is not re-evaluated, where it is the
 * unshifted one.
 * Result: identical bins at a fraction of the arithmetic and 4 bytes per vote. */
#define PC_T22_TURN 4194304u                  /* 2^22 units per turn */
#define PC_T24_SCALE 7680u                    /* 30 bins * 2^8 */
#define PC_T24_MARGIN 64u                     /* in t24 units: 64 * 30 / 2^24 = 1.14e-4 bin (5.7 x the bound) */
#define PC_T24_EDGE (2u * PC_T24_MARGIN * PC_T24_SCALE)   /* position (2^-32 bin) below which a shifted vote is near an edge */
#define PC_T22_FORCE 0xffffffffu              /* "always re-evaluate" marker */
#define PC_T22_PER_RAD 667544.214430109f      /* 2^22 / (2 pi) */

/* (atan2(z, y) + pi) in units of 2^-22 turn, in [0, PC_T22_TURN); PC_T22_FORCE when the
 * vector is zero, not finite or outside 2^-40..2^40, where the products of kernel.cu:84,52
 * could overflow or lose the vector to underflow and the error bound would not hold */
PM_HD uint32_t pc_angle_t22(float y, float z)
{
    const uint32_t by = PM_BITS_F2U(y) & 0x7fffffffu, bz = PM_BITS_F2U(z) & 0x7fffffffu;
    const uint32_t e = (by > bz ? by : bz) >> 23;
    const float q = __builtin_rintf((pm_atan2f(z, y) + PM_PI_F) * PC_T22_PER_RAD);
    if (e < 87u || e > 167u || !(q >= 0.0f && q <= 4194304.0f)) return PC_T22_FORCE;
    return (uint32_t)q & (PC_T22_TURN - 1u);
}

/* what a scene pair contributes to every vote of its reference point: 4 * (theta_v + half a
 * turn) + margin; only the low 24 bits matter */
PM_HD uint32_t pc_vote_base_t24(uint32_t theta_v_t22)
{
    return ((theta_v_t22 + PC_T22_TURN / 2u) << 2) + PC_T24_MARGIN;
}

/* A model pair entry as the vote kernel streams it: theta_u21 << 11 | half << 10 | row.
 *   theta_u21 = the entry's angle in units of 2^-21 turn (pc_angle_t22 rounded to even units: the model side
 *               gives up one bit so that the row field has eleven);
 *   row       = the accumulator row (0..1022; 1023 = the sink row of padding entries);
 *   half      = which 16-bit half of the row's counters the model reference point owns: a table slice holds
 *               2 x 1023 model reference points, two per 32-bit counter word.
 * A vote is tm32 = pc_vote_base_t32(theta_v) - word: the angle difference sits in the upper 24 bits (units of
 * 2^-24 turn << 8), and the eleven row bits ride along below them -- they lower the difference by at most
 * 2047/256 = 8 units of 2^-24 turn, which the base centres (+ 1024 = 4 units).  Error budget against the
 * reference's alpha, in units of 2^-24 turn (one unit = 1.79e-6 bin): 11.2 for everything of DESIGN.md 5 with
 * both angles at 2^-22 turn, + 2 for the coarser model angle, + 4 for the row bits = 17.2; the margin is 64:
 * a vote is re-evaluated whenever it lies within 60 units of a bin edge, 3.5 x the budget (4.5 x without the
 * worst-case stacking of the row bits).  math_exhaustive `hybrid` runs exactly this arithmetic. */
#define PC_ROW_BITS 11
#define PC_ROW_MASK 0x7ffu
#define PC_ROW10_MASK 0x3ffu
#define PC_ROW_HALF_BIT 10
#define PC_ROW_SINK 1023u
#define PC_ROWS_PER_HALF 1023u
PM_HD uint32_t pc_theta_u21(uint32_t theta_t22) { return ((theta_t22 + 1u) >> 1) & (PC_T22_TURN / 2u - 1u); }
PM_HD uint32_t pc_entry_word(uint32_t theta_u_t22, uint32_t row11) { return (pc_theta_u21(theta_u_t22) << PC_ROW_BITS) | row11; }
/* row field of the model reference point with index `local` (0 .. 2*1023-1) inside its slice */
PM_HD uint32_t pc_row11(uint32_t local) { return ((local / PC_ROWS_PER_HALF) << PC_ROW_HALF_BIT) | (local % PC_ROWS_PER_HALF); }
PM_HD uint32_t pc_local_of_row11(uint32_t row11) { return (row11 >> PC_ROW_HALF_BIT) * PC_ROWS_PER_HALF + (row11 & PC_ROW10_MASK); }
PM_HD uint32_t pc_vote_base_t32(uint32_t theta_v_t22) { return (pc_vote_base_t24(theta_v_t22) << 8) + 1024u; }

/* the reference's bin from the stored quantities (host-side statement of what the
 * vote kernel does; used by the CPU check of the scheme) for an entry in accumulator row `row`.
 * *needs_exact (optional) reports whether the vote had to be re-evaluated, *pos_bins the
 * unshifted quantised position. */
PM_HD unsigned pc_alpha_bin_hybrid_ex(float uy, float uz, float vy, float vz, uint32_t row, const uint32_t *tbl,
                                      int *needs_exact, double *pos_bins)
{
    const uint32_t cs = pc_angle_t22(vy, vz), am = pc_angle_t22(uy, uz);
    const uint32_t tm32 = pc_vote_base_t32(cs) - pc_entry_word(am, row);
    const uint64_t prod = (uint64_t)tm32 * 30u;           /* bin * 2^32 + position inside the (shifted) bin */
    const int slow = cs == PC_T22_FORCE || am == PC_T22_FORCE || (uint32_t)prod < PC_T24_EDGE;
    if (needs_exact) *needs_exact = slow;
    if (pos_bins) *pos_bins = (double)((((cs + PC_T22_TURN / 2u) << 2) - (pc_theta_u21(am) << 3)) & 0xffffffu) * 30.0 / 16777216.0;
    if (slow) return pc_alpha_bin_table(uy, uz, vy, vz, tbl);
    return (unsigned)(prod >> 32);
}
PM_HD unsigned pc_alpha_bin_hybrid(float uy, float uz, float vy, float vz, uint32_t row, const uint32_t *tbl)
{
    return pc_alpha_bin_hybrid_ex(uy, uz, vy, vz, row, tbl, 0, 0);
}

#endif /* OSLAM_PPF_CORE_H */
