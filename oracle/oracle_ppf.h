/*
 * oracle_ppf.h -- CPU restatement of the reference's PPF registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under objective-slam_amd/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Pinning status: the reference (CUDA C++, Thrust, PCL, Eigen, Boost) cannot
 * be built in this image without stand-ins for the CUDA device runtime, so
 * there is no oracle/_ref.  The reference tree holds no golden vectors.  The
 * only recorded outputs of the reference itself are the known-answer values
 * in SURVEY.md section 8c (FNV-1a vector; discretised PPFs and keys of a
 * 3-point cloud); tests/test_oracle.py checks this file against all of them.
 * Everything those values do not cover (vote codes, accumulator peaks, pose
 * computation, clustering, ht_dist) is "parity unpinned": restated from the
 * source text, checked only through the geometric properties the reference's
 * own check scripts test (matlab/utils/transform_check.m,
 * translation_vector_processing_check.m).
 *
 * libm: acosf/atan2f/sinf/cosf/fmodf/sqrtf/lrintf are the host libm's, as
 * they would be for the reference's __host__ __device__ code on this host.
 */
#ifndef ORACLE_PPF_H
#define ORACLE_PPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_f3;
typedef struct { float x, y, z, w; } orc_f4;

/* kernel.cu:23-30 (seed default kernel.h:22) */
uint32_t orc_hash(const void *data, int n, uint32_t seed);
/* kernel.cu:109-122, 90-100 */
orc_f4 orc_compute_ppf(orc_f3 p1, orc_f3 n1, orc_f3 p2, orc_f3 n2);
orc_f4 orc_disc_feature(orc_f4 f, float d_dist, float d_angle);
float orc_d_angle0(void);
/* K1 + K2 (kernel.cu:404-477): all-pairs discretised PPFs and keys, row-major
 * [count][count]; rows with idx % df != 0 and the diagonal get NaN / key 0. */
void orc_ppf_all_pairs(const orc_f3 *pts, const orc_f3 *nrm, int count, int df, float d_dist,
                       orc_f4 *ppf_out /* may be NULL */, uint32_t *keys_out);
/* keys of one reference row only (same arithmetic) */
void orc_ppf_row_keys(const orc_f3 *pts, const orc_f3 *nrm, int count, int ref, float d_dist,
                      uint32_t *keys_out);

/* parallel_hash_array.hpp:55-77: sorted unique keys, counts, first index, and
 * the permutation (stable by flat index). Caller frees with orc_table_free. */
typedef struct {
    size_t n;          /* number of data items */
    size_t n_unique;
    uint32_t *keys;    /* [n_unique] sorted */
    size_t *counts;    /* [n_unique] */
    size_t *first;     /* [n_unique] */
    size_t *map;       /* [n] flat pair index, grouped by key */
} orc_table;
void orc_table_build(const uint32_t *keys, size_t n, orc_table *t);
void orc_table_free(orc_table *t);
/* thrust::lower_bound semantics; returns n_unique when key is past the end */
size_t orc_table_lower_bound(const orc_table *t, uint32_t key);

/* kernel.cu:302-349: alpha bin of one (model pair, scene pair) */
unsigned orc_trans_model_scene(orc_f3 m_r, orc_f3 n_r_m, orc_f3 m_i, orc_f3 s_r, orc_f3 n_r_s,
                               orc_f3 s_i, float T_out[16] /* may be NULL */);

/* One accumulator cell of the sparse (s_r, m_r, alpha) histogram */
typedef struct {
    uint64_t code;   /* s_r << 32 | m_r << 6 | alpha_idx  (kernel.cu:549) */
    uint32_t count;
} orc_cell;

typedef struct {
    uint64_t num_scene_ppfs;   /* valid ordered scene pairs (ref r, i != r) */
    uint64_t num_hits;         /* scene pairs whose key is in the model table */
    uint64_t num_votes;        /* num_nonunique_votes, model.cu:122 */
    uint64_t num_unique_votes; /* model.cu:152 */
    uint64_t num_model_keys;   /* num_bins of the model table, util.hpp:46 */
    uint32_t max_count;
    uint64_t num_top;          /* model.cu:168 */
} orc_stats;

/* model.cu:95-171 literally: N^2 keys, K3/K4 vote codes, sort, run-length,
 * sort by (count desc, code asc), keep count > thresh * max.
 * Returns malloc'd cells (caller frees), *n_out = number kept.
 * If all_cells/all_n are non-NULL they receive every unique cell (sorted by
 * code) before thresholding. */
orc_cell *orc_votes_literal(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                            const orc_f3 *sn, int S, int df, float d_dist, float thresh,
                            size_t *n_out, orc_stats *st, orc_cell **all_cells, size_t *all_n);

/* Same result, computed per scene reference point with a dense [M][32]
 * accumulator (nothing N^2-sized); OpenMP over reference points.  ref_begin /
 * ref_step / ref_limit select reference points r = df*(ref_begin + k*ref_step),
 * k < ref_limit (ref_limit < 0: all).  Used for large clouds and as the timed
 * CPU baseline. */
orc_cell *orc_votes_fused(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                          const orc_f3 *sn, int S, int df, float d_dist, float thresh,
                          long ref_begin, long ref_step, long ref_limit, int threads,
                          size_t *n_out, orc_stats *st);
/* the same with the model table built once (CPU-baseline timing excludes the build) */
void *orc_fused_create(const orc_f3 *mp, const orc_f3 *mn, int M, float d_dist);
orc_cell *orc_fused_votes(void *h, const orc_f3 *sp, const orc_f3 *sn, int S, int df, float d_dist,
                          float thresh, long ref_begin, long ref_step, long ref_limit, int threads,
                          size_t *n_out, orc_stats *st);
void orc_fused_free(void *h);
/* dense accumulator [M][32] of one scene reference point (fused path) */
void orc_accumulator_for_ref(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                             const orc_f3 *sn, int S, int s_r, float d_dist, uint32_t *acc);

/* K5 (kernel.cu:352-401,605-645): pose per cell, row-major 4x4 each */
void orc_trans_calc2(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                     const orc_f3 *sp, const orc_f3 *sn, float *T_out);
/* K7 (kernel.cu:124-144,647-661) */
void orc_mat2transquat(const float *T, size_t n, orc_f3 *trans, orc_f4 *quat);
/* K8 (kernel.cu:663-699) */
void orc_trans2idx(const orc_f3 *trans, size_t n, float d_dist, uint32_t *trans_hash,
                   uint32_t *adjacent /* [27*n] */);
/* K6 + K8 + table + K9 + argmax (model.cu:173-244,292-295).
 * trans is updated in place when use_averaged_clusters (as the reference).
 * Returns max_idx; scores_out[n] receives vote_counts_out. */
size_t orc_cluster_gpu_style(const orc_cell *cells, size_t n, orc_f3 *trans, const orc_f4 *quat,
                             float d_dist, int use_l1_norm, int use_averaged_clusters,
                             float *scores_out);
/* transformation_clustering.cpp:62-137 (Eigen parts restated in closed form:
 * parity unpinned). Returns number of result poses (<= 3); T_out[3][16]. */
int orc_cluster_poses_cpu(const float *T, const orc_cell *cells, size_t n, float trans_thresh,
                          float rot_thresh, float *T_out, uint32_t *votes_out);
/* linalg.cu:9-20 */
void orc_ht_dist(const float A[16], const float B[16], float out[2]);

/* ppf_lookup + result extraction (model.cu:269-306, ppf.cu:74-93) on cells
 * already thresholded and ordered.  Returns 0, or 1 when there are no cells. */
int orc_pose_from_cells(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                        const orc_f3 *sp, const orc_f3 *sn, float d_dist, int cpu_clustering,
                        int use_l1_norm, int use_averaged_clusters, float T_out[16]);
/* the same with Model::SetModelPointVoteWeights (model.cu:84-93, kernel.cu:766-782); NULL = all 1 */
int orc_pose_from_cells_w(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                          const orc_f3 *sp, const orc_f3 *sn, float d_dist, int cpu_clustering,
                          int use_l1_norm, int use_averaged_clusters, const float *model_point_weights,
                          float T_out[16]);
size_t orc_cluster_gpu_style_w(const orc_cell *cells, size_t n, orc_f3 *trans, const orc_f4 *quat,
                               float d_dist, int use_l1_norm, int use_averaged_clusters,
                               const float *model_point_weights, float *scores_out);

#ifdef __cplusplus
}
#endif
#endif
