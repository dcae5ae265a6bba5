/*
 * oslam_pose_math.h -- the float sequences of the pose tail, written once for the host C code
 * (oslam_pose.c) and the gfx950 kernels (oslam_posegpu.hip), so that both sides round
 * identically (both are built with -ffp-contract=off).  Reference counterparts
 * (pcl/alignment/src/cuda/kernel.cu): mat4f_mul :211-223, invht :254-299, compute_transforms
 * :372-401, hrotmat2quat :128-144, discretize / trans2idx :102-107,663-699, hash :23-30.
 * The trigonometry (sinf/cosf/atan2f of libm) stays on the host: the frames T_g of the points
 * and the 64 possible rotations about x are computed there and handed to the kernels as tables.
 */
#ifndef OSLAM_POSE_MATH_H
#define OSLAM_POSE_MATH_H

#include <math.h>

#include "ppf_math.h"

PM_HD void pq_mat_zero(float *T)
{
    int i;
    for (i = 0; i < 16; i++) T[i] = 0.0f;
}

PM_HD void pq_mat_mul(const float *A, const float *B, float *C)   /* kernel.cu:211-223 */
{
    int i, j, k;
    pq_mat_zero(C);
    for (i = 0; i < 4; i++)
        for (j = 0; j < 4; j++)
            for (k = 0; k < 4; k++) C[4 * i + j] += A[4 * i + k] * B[4 * k + j];
}

PM_HD void pq_mat_inv_rigid(const float *T, float *I)   /* invht :254-299 */
{
    float nr[9];
    int i, j;
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) {
            I[4 * i + j] = T[4 * j + i];
            nr[3 * i + j] = -I[4 * i + j];
        }
    for (i = 0; i < 3; i++)
        I[4 * i + 3] = nr[3 * i] * T[3] + nr[3 * i + 1] * T[7] + nr[3 * i + 2] * T[11];
    I[12] = 0; I[13] = 0; I[14] = 0; I[15] = 1;
}

/* rotx(theta) from cos and sin of theta (kernel.cu:181-189) */
PM_HD void pq_mat_rotx(float c, float s, float *T)
{
    const float ms = -1 * s;
    pq_mat_zero(T);
    T[15] = 1;
    T[0] = 1; T[5] = c; T[9] = s; T[6] = ms; T[10] = c;
}

/* K5: T = inv(T_s_g) * rotx(alpha) * T_m_g (kernel.cu:372-401); cs = {cos, sin} of alpha_idx*D - pi */
PM_HD void pq_cell_pose(const float *Tm, const float *Ts, float c, float s, float *T)
{
    float rx[16], inv[16], tmp[16];
    pq_mat_rotx(c, s, rx);
    pq_mat_inv_rigid(Ts, inv);
    pq_mat_mul(inv, rx, tmp);
    pq_mat_mul(tmp, Tm, T);
}

/* K7 (kernel.cu:128-144): q = (w,x,y,z) */
PM_HD void pq_pose_quat(const float *T, float q[4])
{
    float t = T[0] + T[5] + T[10];
    float r = pm_sqrtf(1 + t), n;
    q[0] = 0.5f * r;
    q[1] = __builtin_copysignf(0.5f * pm_sqrtf(1 + T[0] - T[5] - T[10]), T[9] - T[6]);
    q[2] = __builtin_copysignf(0.5f * pm_sqrtf(1 - T[0] + T[5] - T[10]), T[2] - T[8]);
    q[3] = __builtin_copysignf(0.5f * pm_sqrtf(1 - T[0] - T[5] + T[10]), T[4] - T[1]);
    n = pm_sqrtf(pm_sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* K8: translation cell (kernel.cu:102-107,674-680) and its FNV key (:23-30 over the 12 bytes) */
PM_HD float pq_quant_down(float x, float y) { return x - fmodf(x, y); }
PM_HD int32_t pq_cell_coord(float t, float d_dist) { return (int32_t)(pq_quant_down(t, d_dist) / d_dist); }
PM_HD uint32_t pq_fnv_cell(const int32_t c[3])
{
    uint32_t h = PM_FNV_BASIS;
    int i;
    for (i = 0; i < 3; i++) h = pm_fnv1a_word(h, (uint32_t)c[i]);
    return h;
}

#endif
