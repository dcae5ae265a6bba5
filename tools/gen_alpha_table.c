/*
 * gen_alpha_table.c -- derives the threshold table behind pc_alpha_bin_table().
 *
 * For a vote with cross = cross(u,v).x and dot = dot(u,v) (reference
 * kernel.cu:338-342) the alpha bin is floor((atan2f(cross, dot) + pi_f) / D).
 * In atan2f's main path the result depends only on q = fl(|cross/dot|) and the
 * two signs (quadrant m = sign(cross) | sign(dot) << 1), and the bin is a
 * monotone step function of q.  This program evaluates the exact bin for EVERY
 * float q in [2^-63, 2^63] and every quadrant, checks monotonicity, and prints
 * the q values at which the bin steps as a C header.  tests/test_math_exact.py
 * re-runs it in --check mode against the committed header.
 *
 *   gen_alpha_table          > ppf_alpha_table.h
 *   gen_alpha_table --check    (exit 0 iff the committed table reproduces every bin)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ppf_math.h"
#ifdef CHECK_HEADER
#include "ppf_alpha_table.h"
#endif

#define Q_LO 0x20000000u   /* 2^-63 */
#define Q_HI 0x5f000000u   /* 2^63  */

/* bin of the main path of pm_atan2f for quadrant m and ratio q */
static int bin_of(uint32_t qbits, int m)
{
    const float pi = PM_BITS_U2F(0x40490fdbu), pi_lo = PM_BITS_U2F(0xb3bbbd2eu);
    float z = pm_atanf_pos_(PM_BITS_U2F(qbits)), alpha, zl = z - pi_lo;
    int k;
    switch (m) {
    case 0: alpha = z; break;
    case 1: alpha = -z; break;
    case 2: alpha = pi - zl; break;
    default: alpha = zl - pi; break;
    }
    (void)pm_quant_down_pos(alpha + PM_PI_F, PM_D_ANGLE, 1.0f / PM_D_ANGLE, &k);
    return k;
}

typedef struct { uint32_t q; int from, to; } step_t;

int main(int argc, char **argv)
{
    int check = argc > 1 && !strcmp(argv[1], "--check");
    enum { NCH = 4096 };
    static step_t steps[4][64];
    int nsteps[4] = {0, 0, 0, 0}, base[4];
    uint64_t span = (uint64_t)Q_HI - Q_LO + 1, per = (span + NCH - 1) / NCH;
    int bad = 0;

    for (int m = 0; m < 4; m++) {
        static step_t chunk_steps[NCH][8];
        static int chunk_n[NCH], chunk_first[NCH], chunk_last[NCH];
#pragma omp parallel for schedule(dynamic, 8)
        for (int c = 0; c < NCH; c++) {
            uint64_t b = Q_LO + (uint64_t)c * per, e = b + per;
            if (e > (uint64_t)Q_HI + 1) e = (uint64_t)Q_HI + 1;
            int prev = bin_of((uint32_t)b, m), n = 0;
            chunk_first[c] = prev;
            for (uint64_t q = b + 1; q < e; q++) {
                int k = bin_of((uint32_t)q, m);
                if (k != prev) {
                    if (n < 8) { chunk_steps[c][n].q = (uint32_t)q; chunk_steps[c][n].from = prev; chunk_steps[c][n].to = k; }
                    n++;
                    prev = k;
                }
            }
            chunk_n[c] = n;
            chunk_last[c] = prev;
        }
        base[m] = chunk_first[0];
        int prev = chunk_first[0];
        for (int c = 0; c < NCH; c++) {
            if (chunk_n[c] > 8) { fprintf(stderr, "too many steps in a chunk\n"); return 2; }
            if (chunk_first[c] != prev) {   /* a step exactly at the chunk boundary */
                steps[m][nsteps[m]].q = (uint32_t)(Q_LO + (uint64_t)c * per);
                steps[m][nsteps[m]].from = prev;
                steps[m][nsteps[m]].to = chunk_first[c];
                nsteps[m]++;
            }
            for (int i = 0; i < chunk_n[c]; i++) steps[m][nsteps[m]++] = chunk_steps[c][i];
            prev = chunk_last[c];
        }
        int dir = (m == 0 || m == 3) ? 1 : -1;
        for (int i = 0; i < nsteps[m]; i++)
            if (steps[m][i].to - steps[m][i].from != dir) { fprintf(stderr, "quadrant %d: non-monotone step at %08x\n", m, steps[m][i].q); bad = 1; }
        if (nsteps[m] > 8) { fprintf(stderr, "quadrant %d: %d steps > 8\n", m, nsteps[m]); bad = 1; }
    }
    if (bad) return 1;

    /* row layout: [0] = the one tiny threshold of quadrants 1 and 2 (bin 15 -> 14, 30 -> 29
     * as soon as z > 0), or +inf; [1..7] = the seven thresholds near tan(12 deg * j) */
    uint32_t row[4][8];
    for (int m = 0; m < 4; m++) {
        int n = 0;
        row[m][0] = 0x7f800000u;
        for (int i = 0; i < nsteps[m]; i++) {
            if (steps[m][i].q < 0x3a800000u) {          /* < 2^-10 */
                if (i != 0) { fprintf(stderr, "unexpected tiny threshold order\n"); return 1; }
                row[m][0] = steps[m][i].q;
            } else {
                if (n >= 7) { fprintf(stderr, "quadrant %d: more than 7 main thresholds\n", m); return 1; }
                row[m][1 + n++] = steps[m][i].q;
            }
        }
        if (n != 7) { fprintf(stderr, "quadrant %d: %d main thresholds\n", m, n); return 1; }
    }

    if (check) {
#ifdef CHECK_HEADER
        for (int m = 0; m < 4; m++) {
            if (PC_ALPHA_BASE[m] != base[m]) { fprintf(stderr, "base mismatch\n"); return 1; }
            for (int i = 0; i < 8; i++)
                if (PC_ALPHA_THR[m][i] != row[m][i]) { fprintf(stderr, "threshold mismatch m=%d i=%d\n", m, i); return 1; }
        }
        printf("alpha table verified on every q in [2^-63, 2^63] x 4 quadrants\n");
        return 0;
#else
        fprintf(stderr, "built without CHECK_HEADER\n");
        return 2;
#endif
    }

    printf("/* Generated by tools/gen_alpha_table.c from an exhaustive sweep of every float ratio\n"
           " * q in [2^-63, 2^63]: do not edit.\n"
           " * bin = PC_ALPHA_BASE[m] +/- #{i : q >= PC_ALPHA_THR[m][i]}  (+ for quadrants 0 and 3,\n"
           " * - for 1 and 2), quadrant m = sign(cross) | sign(dot) << 1.  Thresholds are float bit\n"
           " * patterns; row[0] is the tiny threshold of quadrants 1 and 2 (or +inf), row[1..7]\n"
           " * ascend and sit near tan(12 deg * j). */\n"
           "#ifndef OSLAM_PPF_ALPHA_TABLE_H\n#define OSLAM_PPF_ALPHA_TABLE_H\n#include <stdint.h>\n");
    printf("static const int PC_ALPHA_BASE[4] = {%d, %d, %d, %d};\n", base[0], base[1], base[2], base[3]);
    printf("static const uint32_t PC_ALPHA_THR[4][8] = {\n");
    for (int m = 0; m < 4; m++) {
        printf("    {");
        for (int i = 0; i < 8; i++) printf("0x%08xu%s", row[m][i], i < 7 ? ", " : "");
        printf("},\n");
    }
    printf("};\n/* the same 32 words as an initialiser list (device copy) */\n#define PC_ALPHA_THR_FLAT \\\n");
    for (int m = 0; m < 4; m++) {
        printf("    ");
        for (int i = 0; i < 8; i++) printf("0x%08xu%s", row[m][i], (m == 3 && i == 7) ? "" : ", ");
        printf("%s\n", m == 3 ? "" : " \\");
    }
    printf("#endif\n");
    return 0;
}
