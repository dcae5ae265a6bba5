/*
 * Exhaustive / sampled comparison of ppf_math.h against the host libm.
 * Test infrastructure only.
 *
 *   math_exhaustive acosf  STRIDE   all floats i*STRIDE: pm_acosf vs acosf
 *   math_exhaustive atanf  STRIDE   all floats i*STRIDE: pm_atanf vs atanf
 *   math_exhaustive atan2f N        N pseudo-random + structured (y,x) pairs
 *   math_exhaustive quant  N        pm_quant_down_pos vs x - fmodf(x, step)
 *   math_exhaustive alphabin N      pc_alpha_bin_table vs the libm formula of kernel.cu:338-342
 *   math_exhaustive hybrid N        pc_alpha_bin_hybrid vs the same formula; also prints the largest
 *                                   distance between the quantised and the reference position
 *   math_exhaustive acosbin STRIDE  all floats i*STRIDE: pc_acos_bin (the tabulated steps) vs the quantised libm
 *                                   acosf, acosf(c) - fmodf(acosf(c), D) == bin * D (or NaN, bin 16)
 *   math_exhaustive pairbins N      N point pairs (random, near-degenerate and degenerate): the key rebuilt from
 *                                   pc_pair_bins by pc_key_of_bins vs pc_pair_key
 *
 * Prints "mismatches=K checked=N" and exits non-zero when K != 0.
 * NaN results compare equal when both are NaN (payload is canonicalised by the
 * caller of pm_acosf; see ppf_math.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ppf_core.h"

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static int cs_ok(float y, float z) { return pc_angle_t22(y, z) != PC_T22_FORCE; }

static int same(float a, float b)
{
    if (isnan(a) && isnan(b)) return 1;
    return PM_BITS_F2U(a) == PM_BITS_F2U(b);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const char *mode = argv[1];
    uint64_t arg = strtoull(argv[2], 0, 10);
    uint64_t bad = 0, checked = 0;

    if (!strcmp(mode, "acosf") || !strcmp(mode, "atanf")) {
        int is_acos = !strcmp(mode, "acosf");
        uint64_t n = (1ull << 32) / arg;
#pragma omp parallel for reduction(+ : bad, checked) schedule(static)
        for (uint64_t i = 0; i < n; i++) {
            float x = PM_BITS_U2F((uint32_t)(i * arg));
            float a = is_acos ? pm_acosf(x) : pm_atanf(x);
            float b = is_acos ? acosf(x) : atanf(x);
            checked++;
            if (!same(a, b)) {
                bad++;
                if (bad < 5) fprintf(stderr, "%s(%a): pm=%a libm=%a\n", mode, x, a, b);
            }
        }
    } else if (!strcmp(mode, "atan2f")) {
#pragma omp parallel for reduction(+ : bad, checked) schedule(static)
        for (uint64_t i = 0; i < arg; i++) {
            uint64_t s = i * 0x2545f4914f6cdd1dull + 12345;
            uint64_t r = splitmix(&s);
            float y, x;
            switch (i & 3) {
            case 0: /* arbitrary bit patterns */
                y = PM_BITS_U2F((uint32_t)r);
                x = PM_BITS_U2F((uint32_t)(r >> 32));
                break;
            case 1: { /* moderate magnitudes, the regime the vote path lives in */
                y = (float)((double)(int32_t)(uint32_t)r / 2147483648.0 * 4.0);
                x = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0 * 4.0);
                break;
            }
            case 2: { /* ratios near the atanf range-reduction thresholds */
                static const float thr[5] = {0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.0f};
                x = (float)((double)(uint32_t)r / 4294967296.0 + 0.01);
                if (r >> 63) x = -x;
                y = x * thr[(r >> 40) % 5];
                y = PM_BITS_U2F(PM_BITS_F2U(y) + (uint32_t)((r >> 32) & 7) - 3);
                if ((r >> 62) & 1) y = -y;
                break;
            }
            default: { /* zeros, infinities, huge exponent gaps */
                static const uint32_t sp[8] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u,
                                               0x3f800000u, 0x00000001u, 0x7f7fffffu, 0x5d5e0b6bu};
                y = PM_BITS_U2F(sp[r & 7]);
                x = PM_BITS_U2F(((r >> 8) & 1) ? sp[(r >> 3) & 7] : (uint32_t)(r >> 32));
                break;
            }
            }
            float a = pm_atan2f(y, x), b = atan2f(y, x);
            checked++;
            if (!same(a, b)) {
                bad++;
                if (bad < 5) fprintf(stderr, "atan2f(%a,%a): pm=%a libm=%a\n", y, x, a, b);
            }
        }
    } else if (!strcmp(mode, "quant")) {
#pragma omp parallel for reduction(+ : bad, checked) schedule(static)
        for (uint64_t i = 0; i < arg; i++) {
            uint64_t s = i * 0x9e3779b97f4a7c15ull + 777;
            uint64_t r = splitmix(&s);
            float step, x;
            int k;
            if (i & 1) {
                step = PM_D_ANGLE;
                x = (float)((double)(uint32_t)r / 4294967296.0 * 6.2831856);
            } else {
                step = (float)((double)(uint32_t)(r >> 32) / 4294967296.0 * 0.3 + 1e-3);
                x = (float)((double)(uint32_t)r / 4294967296.0) * step * (float)((r >> 20) % 3000);
            }
            if ((i % 1000) == 0) { /* exact multiples: the boundary cases */
                x = step * (float)((r >> 8) % 64);
            }
            float inv = 1.0f / step;
            float a = pm_quant_down_pos(x, step, inv, &k);
            float b = x - fmodf(x, step);
            checked++;
            if (k < 0) continue;
            if (!same(a, b)) {
                bad++;
                if (bad < 5) fprintf(stderr, "quant(%a,%a): pm=%a libm=%a\n", x, step, a, b);
            }
        }
    } else if (!strcmp(mode, "acosbin")) {
        uint64_t n = (1ull << 32) / arg;
#pragma omp parallel for reduction(+ : bad, checked) schedule(static)
        for (uint64_t i = 0; i < n; i++) {
            const float c = PM_BITS_U2F((uint32_t)(i * arg));
            const uint32_t b = pc_acos_bin(c, PC_ACOS_LUT);
            const float a = acosf(c), q = a - fmodf(a, PM_D_ANGLE);
            const int ok = b == 16u ? isnan(q) : (!isnan(q) && PM_BITS_F2U(q) == PM_BITS_F2U((float)b * PM_D_ANGLE));
            checked++;
            if (!ok) {
                bad++;
                if (bad < 5) fprintf(stderr, "acosbin(%a): table=%u libm quantised=%a\n", c, b, q);
            }
        }
    } else if (!strcmp(mode, "pairbins")) {
#pragma omp parallel for reduction(+ : bad, checked) schedule(static)
        for (uint64_t i = 0; i < arg; i++) {
            uint64_t s = i * 0x9e3779b97f4a7c15ull + 4242;
            float v[12];
            for (int j = 0; j < 12; j++) v[j] = (float)((double)(uint32_t)splitmix(&s) / 4294967296.0 * 2.0 - 1.0);
            const uint64_t r = splitmix(&s);
            /* the special cases the kernels meet: axis-aligned and parallel normals (acos arguments of exactly
             * 0 and +-1), a zero normal, coincident points, a non-finite coordinate, lattice points */
            switch (r % 16) {
            case 0: v[3] = 0; v[4] = 0; v[5] = 1; v[9] = 0; v[10] = 0; v[11] = 1; break;
            case 1: v[9] = v[3]; v[10] = v[4]; v[11] = v[5]; break;
            case 2: v[9] = -v[3]; v[10] = -v[4]; v[11] = -v[5]; break;
            case 3: v[3] = v[4] = v[5] = 0; break;
            case 4: v[6] = v[0]; v[7] = v[1]; v[8] = v[2]; break;
            case 5: v[(r >> 8) % 12] = (r >> 16) & 1 ? NAN : INFINITY; break;
            case 6: for (int j = 0; j < 12; j++) v[j] = (float)(int)(v[j] * 4.0f) * 0.25f; break;
            case 7: v[2] = v[8] = 0; v[3] = v[4] = v[9] = v[10] = 0; v[5] = v[11] = 1; break;
            case 8: v[6] = v[0] + v[3]; v[7] = v[1] + v[4]; v[8] = v[2] + v[5]; break;
            default: break;
            }
            const float d_dist = (r >> 24) % 3 == 0 ? 0.025f : (float)((double)(uint32_t)(r >> 32) / 4294967296.0 * 0.2 + 1e-3);
            const float inv = 1.0f / d_dist;
            const float n1n = pc_norm3(v[3], v[4], v[5]), n2n = pc_norm3(v[9], v[10], v[11]);
            const uint32_t key = pc_pair_key(v[0], v[1], v[2], v[3], v[4], v[5], n1n, v[6], v[7], v[8], v[9], v[10], v[11], n2n, d_dist, inv);
            uint32_t combo;
            const int k1 = pc_pair_bins(v[0], v[1], v[2], v[3], v[4], v[5], n1n, v[6], v[7], v[8], v[9], v[10], v[11], n2n, d_dist, inv,
                                        PC_ACOS_LUT, &combo);
            checked++;
            if (k1 < 0) continue;                 /* generic path: the kernels call pc_pair_key */
            if (combo >= PC_ANGLE_COMBOS || pc_key_of_bins((uint32_t)k1, combo, d_dist) != key) {
                bad++;
                if (bad < 5) fprintf(stderr, "pairbins case %llu: key %08x, bins give %08x\n", (unsigned long long)i, key,
                                     pc_key_of_bins((uint32_t)k1, combo, d_dist));
            }
        }
    } else if (!strcmp(mode, "alphabin") || !strcmp(mode, "hybrid")) {
        const int hybrid = !strcmp(mode, "hybrid");
        double worst = 0.0;
        uint64_t slow = 0;
#pragma omp parallel for reduction(+ : bad, checked, slow) reduction(max : worst) schedule(static)
        for (uint64_t i = 0; i < arg; i++) {
            uint64_t s = i * 0xd1342543de82ef95ull + 99;
            uint64_t r = splitmix(&s), r2 = splitmix(&s);
            float uy, uz, vy, vz;
            if ((i & 7) == 7) {          /* arbitrary bit patterns */
                uy = PM_BITS_U2F((uint32_t)r); uz = PM_BITS_U2F((uint32_t)(r >> 32));
                vy = PM_BITS_U2F((uint32_t)r2); vz = PM_BITS_U2F((uint32_t)(r2 >> 32));
            } else {
                uy = (float)((double)(int32_t)(uint32_t)r / 2147483648.0 * 3.0);
                uz = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0 * 3.0);
                vy = (float)((double)(int32_t)(uint32_t)r2 / 2147483648.0 * 3.0);
                vz = (float)((double)(int32_t)(uint32_t)(r2 >> 32) / 2147483648.0 * 3.0);
                if ((i & 7) == 1) { vy = uy; vz = uz; }              /* alpha = 0 */
                if ((i & 7) == 2) { vy = -uy; vz = -uz; }            /* alpha = +-pi */
                if ((i & 7) == 3) { vy = -uz; vz = uy; }             /* alpha = pi/2 */
                if ((i & 7) == 5) {                                  /* magnitudes from 2^-50 to 2^50 */
                    float su = ldexpf(1.0f, (int)((r2 >> 3) % 101) - 50), sv = ldexpf(1.0f, (int)((r2 >> 13) % 101) - 50);
                    uy *= su; uz *= su; vy *= sv; vz *= sv;
                }
                if ((i & 7) == 4) {                                  /* on a bin edge: alpha = 12 deg * j */
                    double a = 0.20943951023931953 * (double)((r2 >> 7) % 30), c = cos(a), sn = sin(a);
                    vy = (float)(uy * c - uz * sn); vz = (float)(uy * sn + uz * c);
                }
            }
            /* the row bits of the entry word ride along in the subtraction: both ends of their range must agree */
            unsigned a = hybrid ? pc_alpha_bin_hybrid(uy, uz, vy, vz, (i & 8) ? PC_ROW_MASK : 0u, &PC_ALPHA_THR[0][0])
                                : pc_alpha_bin_table(uy, uz, vy, vz, &PC_ALPHA_THR[0][0]);
            if (hybrid && a != pc_alpha_bin_hybrid(uy, uz, vy, vz, (uint32_t)(r2 >> 20) & PC_ROW_MASK, &PC_ALPHA_THR[0][0])) a = 254u;
            /* the reference's own sequence with libm (kernel.cu:338-342) */
            float cx = uy * vz - uz * vy, dt = 0.0f * 0.0f + uy * vy + uz * vz;
            float al = atan2f(cx, dt) + PM_PI_F;
            float aq = al - fmodf(al, PM_D_ANGLE);
            unsigned b = isnan(aq) ? 255u : (unsigned)lrintf(aq / PM_D_ANGLE);
            checked++;
            if (hybrid && b != 255u && (i & 7) != 7 && cs_ok(uy, uz) && cs_ok(vy, vz)) {
                /* distance (in bins, on the circle) between the quantised position and the
                 * reference's alpha + pi: the quantity the margin has to cover */
                int ne = 0;
                double pos = 0.0;
                (void)pc_alpha_bin_hybrid_ex(uy, uz, vy, vz, 0u, &PC_ALPHA_THR[0][0], &ne, &pos);
                double d = fabs(pos - (double)al / (double)PM_D_ANGLE);
                if (d > 15.0) d = fabs(d - 30.0);
                if (d > worst) worst = d;
                slow += (uint64_t)ne;
            }
            if (a != b) {
                bad++;
                if (bad < 5) fprintf(stderr, "alphabin(%a,%a,%a,%a): table=%u libm=%u\n", uy, uz, vy, vz, a, b);
            }
        }
        if (hybrid) printf("largest |quantised - reference| = %.3g bin (margin %.3g); re-evaluated %.4f%%\n", worst,
                           (double)PC_T24_MARGIN * 30.0 / 16777216.0, 100.0 * (double)slow / (double)checked);
    } else {
        return 2;
    }
    printf("mismatches=%llu checked=%llu\n", (unsigned long long)bad, (unsigned long long)checked);
    return bad != 0;
}
