/*
 * oslam_pose.h -- host-side tail of the path: accumulator peaks -> poses ->
 * clustering -> best pose.  The north-star keeps this stage on the host, after
 * the all-gather of per-GPU peaks; it touches a few thousand records.
 * Internal to liboslam_hip.so (the exported wrappers are in include/oslam.h).
 */
#ifndef OSLAM_POSE_H
#define OSLAM_POSE_H

#include <stddef.h>
#include <stdint.h>

#include "oslam.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rows y and z of T_g (8 floats) for n reference points: what the kernels need */
void oslam_T_g_rows(const float *xyz, const float *nrm, const uint32_t *idx, size_t n,
                    float *rows_out);

/* the whole frame T_g (16 floats, row-major) of the points idx0, idx0 + step, ... (n of them) */
void oslam_T_g_full(const float *xyz, const float *nrm, size_t idx0, size_t step, size_t n, float *out16);

/* cos, sin of alpha_idx * D - pi for alpha_idx = 0..63 (libm): the kernels' rotation table */
void oslam_rotx_table(float cs[128]);

/* Optional accelerator for the clustering scores (set by oslam_host.c while a device is bound):
 * fills score[n] exactly as the host loop would, returns 0 on success.  hash_idx = n pairs
 * {cell hash, pose index} ascending. */
typedef int (*oslam_cluster_hook)(size_t n, const float *trans, const float *quat, const float *wv,
                                  const int32_t *cell, const uint32_t *hash_idx, float d_dist, int use_l1,
                                  float *score);
void oslam_pose_set_cluster_hook(oslam_cluster_hook hook);

#ifdef __cplusplus
}
#endif
#endif
