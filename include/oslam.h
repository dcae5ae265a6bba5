/*
 * oslam.h -- C-ABI of the MI355X-native PPF registration path
 * (liboslam_hip.so).  Plain C: opaque handles, plain pointers and sizes, every
 * function returns an int status (0 = OSLAM_OK).  Nothing here exits the
 * process (the reference's HANDLE_ERROR does, include/impl/util.hpp:18-26).
 *
 * Each entry point names the reference interface it replaces; paths are
 * relative to the reference's pcl/alignment/.  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Clouds are passed as two float pointers (first coordinate of the first
 * point, first component of the first normal) plus a byte stride between
 * consecutive points, so a pcl::PointCloud<pcl::PointNormal> (48-byte points,
 * xyz at +0, normal at +16) is passed without a copy:
 *     xyz = &cloud[0].x, nrm = &cloud[0].normal_x, stride_bytes = 48.
 * Tightly packed float[n][3] arrays use stride_bytes = 12.  Host pointers.
 */
#ifndef OSLAM_H
#define OSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSLAM_OK 0
#define OSLAM_E_INVALID 1     /* bad argument (NULL, n == 0, d_dist <= 0, mismatching d_dist ...) */
#define OSLAM_E_DEVICE 2      /* HIP error; see oslam_last_error() */
#define OSLAM_E_NOMEM 3
#define OSLAM_E_NO_VOTES 4    /* no scene pair matched the model: T is all zeros */
#define OSLAM_E_LIMIT 5       /* cloud exceeds an encoding limit (see oslam_model_create) */
#define OSLAM_E_PEER 6        /* multi-GPU: another rank failed; every rank abandoned the registration together */

/* Per-vote arithmetic of the alpha angle (reference src/cuda/kernel.cu:302-342). */
#define OSLAM_VOTE_EXACT 0    /* accumulator identical to the reference's: alpha from quantised angles,
                               * re-evaluated with the reference's float sequence near bin edges */
#define OSLAM_VOTE_FAST 1     /* quantised angles only (Drost's alpha_scene - alpha_model): bins can
                               * differ from the reference's within 2e-5 bin of an edge */

/* Flags of the reference's CLI that reach the path (src/alignment.cpp:119-172)
 * plus this build's extensions.  oslam_params_default() fills the reference's
 * defaults. */
typedef struct oslam_params {
    unsigned ref_point_df;         /* --ref_point_df, default 1 */
    float vote_count_threshold;    /* --vote_count_threshold, default 0.4 */
    int cpu_clustering;            /* --cpu_clustering, default 0 */
    int use_l1_norm;               /* --use_l1_norm, default 0 */
    int use_averaged_clusters;     /* --use_averaged_clusters, default 0 */
    int dev;                       /* --dev: device = min(numDevices-1, dev) (src/cuda/ppf.cu:45); default 0 */
    /* extensions */
    int vote_mode;                 /* OSLAM_VOTE_EXACT (default) or OSLAM_VOTE_FAST */
    int shard_rank;                /* scene reference points r = df*(rank + world*t); default 0 */
    int shard_world;               /* default 1 */
    unsigned max_cells;            /* initial capacity of the peak-record buffer, default 1<<22; it grows (to at most
                                    * 2^28 records) when more cells than that lie above the threshold */
    unsigned pose_gpu_min;         /* peak records from which the pose tail runs on the device; 0 = default (4096).  Both
                                    * tails give identical results; tests force either one */
    int no_bucket_spread;          /* model build: skip the pass that orders every bucket for the LDS banks (A/B
                                    * measurements; the accumulators do not depend on the order) */
    unsigned scratch_gib;          /* limit of the device's hit-list pool in GiB; 0 = default (4).  A registration
                                    * whose lists need more runs in batches of reference points */
    int pose_two_sorts;            /* device pose tail: order the kept cells with two stable sorts of (code, count) pairs even
                                    * when their fields fit one packed 64-bit key (the path clouds of 2^28 points with
                                    * 2^32 votes per cell take; identical results, tests force it) */
    int reserved[2];
} oslam_params;

/* Counters the reference logs at debug level (model.cu:122,152,161-168;
 * util.hpp:46) plus timings; all exact integers. */
typedef struct oslam_stats {
    uint64_t num_scene_ppfs;       /* valid ordered scene pairs (reference point r, i != r) on this shard */
    uint64_t num_hits;             /* of those, pairs whose key is in the model table */
    uint64_t num_votes;            /* accumulator increments = num_nonunique_votes */
    uint64_t num_unique_votes;     /* non-empty accumulator cells */
    uint64_t num_model_keys;       /* num_bins of the model table (counts the key-0 self-pair bucket) */
    uint64_t num_top;              /* cells with count > threshold * max */
    uint32_t max_count;            /* largest cell */
    uint32_t num_emitted;          /* records the vote kernel wrote before the final filter */
    float ms_vote;                 /* scene-key, hit-sort and vote kernels of the call, HIP events on the launch stream */
    float ms_total;                /* whole oslam_align call, host clock */
    uint32_t vote_launches;        /* launches of the vote kernel (one per batch of reference points) */
    float ms_vote_kernel;          /* sum over the vote-kernel launches alone (HIP events around each) */
    float ms_key_kernel;           /* sum over the scene-key and hit-sort kernel launches */
    uint32_t wide_workgroups;      /* vote workgroups (reference point x table slice) whose 16-bit counters overflowed and
                                    * were voted again with 32-bit counters (0 unless both clouds hold large planes) */
    uint64_t num_pairs_probed;     /* of num_scene_ppfs, the pairs whose distance bin can reach a model key: they are keyed
                                    * and probed; the others cannot hit and are dropped by the distance test alone */
    uint64_t scratch_bytes;        /* size of the device's hit-list pool after this call */
    uint64_t num_entries_streamed; /* model pair entries the vote kernel read: bucket lengths summed over (run of hits, table slice) */
    uint64_t num_items;            /* (run of hits, table slice) pairs with a bucket = buckets streamed */
} oslam_stats;

/* One accumulator peak: code = s_r << 32 | m_r << 6 | alpha_idx (kernel.cu:549). */
typedef struct oslam_cell {
    uint64_t code;
    uint32_t count;
    uint32_t pad;
} oslam_cell;

typedef struct oslam_model oslam_model;
typedef struct oslam_scene oslam_scene;

/* Reference defaults (src/alignment.cpp:119-172). */
int oslam_params_default(oslam_params *p);

/* d_dist = tau_d * max bounding-box extent (src/alignment.cpp:246-253). */
int oslam_d_dist_from_cloud(const float *xyz, size_t n, size_t stride_bytes, float tau_d,
                            float *d_dist_out);

/* Model::Model (include/model.h:17-19, src/cuda/model.cu:43-82): uploads the
 * cloud, computes all M*(M-1) pair features and builds the HBM-resident hash
 * table.  Limits: 2 <= n <= 46340 (the reference's own 32-bit pair index,
 * kernel.cu:433).  params may be NULL (defaults). */
int oslam_model_create(const float *xyz, const float *nrm, size_t n, size_t stride_bytes,
                       float d_dist, const oslam_params *params, oslam_model **out);
void oslam_model_destroy(oslam_model *m);

/* Persistent model database (the reference rebuilds every model for every scene, src/cuda/ppf.cu:57-70;
 * its own comment at :64-66 asks for this): oslam_model_save writes the built table -- cloud, slice
 * tables, union table, reachable-distance bitset, pair entries, point weights -- to one file;
 * oslam_model_load maps it back into HBM without recomputing a pair.  The file is tied to the
 * table layout version and to the vote mode it was built with (a fast-mode table has no exact
 * entries); a mismatch or a damaged file is OSLAM_E_INVALID.  params (may be NULL) supplies the
 * run-time flags of the loaded model (device, clustering flags, threshold); the table fields of
 * the file win. */
int oslam_model_save(const oslam_model *m, const char *path);
int oslam_model_load(const char *path, const oslam_params *params, oslam_model **out);
/* size of a built model: points, d_dist and the bytes its table holds in HBM (any pointer may be NULL) */
int oslam_model_info(const oslam_model *m, size_t *n_points, float *d_dist, uint64_t *table_bytes);

/* Model::SetModelPointVoteWeights (include/model.h:22); weights[n], default all 1. */
int oslam_model_set_point_weights(oslam_model *m, const float *weights, size_t n);

/* Scene::Scene (include/scene.h:15-16, src/cuda/scene.cu:24-55).  d_dist must
 * equal the d_dist of the model it is aligned with (src/cuda/ppf.cu:64-67), or be 0: the
 * reference discretises the scene's pair features with d_dist when the Scene is built, here the
 * scene holds points and reference frames only, so one scene with d_dist 0 serves a database of
 * models with different d_dist (the pair keys are made per model inside oslam_align). */
int oslam_scene_create(const float *xyz, const float *nrm, size_t n, size_t stride_bytes,
                       float d_dist, unsigned ref_point_downsample_factor,
                       const oslam_params *params, oslam_scene **out);
void oslam_scene_destroy(oslam_scene *s);

/* Model::ppf_lookup + result extraction (model.cu:269-306, ppf.cu:74-93):
 * T_rowmajor receives the best model->scene pose.  stats may be NULL. */
int oslam_align(oslam_model *m, oslam_scene *s, float T_rowmajor[16], oslam_stats *stats);

/* Optional: everything oslam_align would allocate on first use for this pair (the scratch pool of the
 * device -- mapping 32 GiB takes about a second -- and the frame tables of the device pose tail), done
 * ahead of time so that the first registration is as fast as the following ones. */
int oslam_align_prepare(oslam_model *m, oslam_scene *s);

/* Model database: what the loop of src/cuda/ppf.cu:57-100 becomes when the models stay resident.  The
 * database borrows the models (destroy it before them).  Models that share d_dist, device and vote mode
 * form a group with one union table of pair keys: per frame the scene pass (pair keys, probe, hit sort)
 * runs once per group instead of once per model, and every member votes from the same hit lists with
 * its own buckets.  Models with a d_dist of their own are groups of one.  A database made with ONE
 * d_dist for all models (the scene then also needs one voxel grid only) gets the whole benefit.
 * oslam_db_align: T_out[j*16..] = pose of model j (zeros when nothing matched), stats[j] (may be NULL)
 * its counters; in a group of several, num_hits counts the pairs whose key is in ANY member and the
 * kernel times are the group's, shared out evenly. */
typedef struct oslam_db oslam_db;
int oslam_db_create(oslam_model *const *models, size_t n, oslam_db **out);
void oslam_db_destroy(oslam_db *db);
/* the same, for a caller that destroys the models right afterwards: the members do not get their own key
 * tables back (they cannot be aligned any more, only destroyed) */
void oslam_db_destroy_with_models(oslam_db *db);
int oslam_db_align(oslam_db *db, oslam_scene *s, float *T_out, oslam_stats *stats);
int oslam_db_size(const oslam_db *db, size_t *n_models, size_t *n_groups);

/* ppf_registration (include/ppf.h:9-15, src/cuda/ppf.cu:29-106): every scene
 * against every model; T_out[(i*n_models + j)*16 ..] = pose of model j in
 * scene i.  model_weights is accepted and ignored, as in the reference
 * (ppf.cu:35).  Unlike the reference this does not reset the device. */
int oslam_ppf_registration(const float *const *scene_xyz, const float *const *scene_nrm,
                           const size_t *scene_n, size_t n_scenes, const float *const *model_xyz,
                           const float *const *model_nrm, const size_t *model_n, size_t n_models,
                           size_t stride_bytes, const float *model_d_dists,
                           unsigned ref_point_downsample_factor, float vote_count_threshold,
                           int cpu_clustering, int use_l1_norm, int use_averaged_clusters, int devUse,
                           const float *model_weights, float *T_out);

/* ht_dist (include/linalg.h:7, src/cuda/linalg.cu:9-20): out = {|dt|, |angle|}. */
int oslam_ht_dist(const float A[16], const float B[16], float out[2]);

/* voxelGridDownsample (src/alignment.cpp:79-87, applied to scenes with leaf = scene_leaf_size
 * and to models with leaf = d_dist, :265-288; pcl/voxel_grid/voxel_grid.cpp:18-21): one output
 * point per occupied voxel = mean of the points (and of their normals, not renormalised) in
 * it, in ascending voxel index -- pcl::VoxelGrid's algorithm with the point order inside a
 * voxel fixed to the input order.  xyz_out / nrm_out: packed float[cap][3]; *n_out receives the
 * number of voxels (OSLAM_E_LIMIT if it exceeds cap or the voxel count overflows int32). */
int oslam_voxel_grid(const float *xyz, const float *nrm, size_t n, size_t stride_bytes, float leaf,
                     int dev, float *xyz_out, float *nrm_out, size_t cap, size_t *n_out);

/* Depth image -> scene cloud with normals: the front end of a streaming configuration (a range camera
 * feeding the PPF path; the reference takes finished clouds from KinFu, README.md:5-8, and has no code
 * for this step -- the specification is oracle/oracle_depth.c).  Pinhole camera, z along the optical
 * axis: z = raw * depth_scale, valid in [z_min, z_max]; normals from the four axis neighbours (all
 * valid and within max_jump of z), unit length, facing the camera.  Pixels without a normal are
 * dropped; the rest come out in row-major pixel order.  depth: host image, uint16 (depth_is_u16 != 0)
 * or float, width x height.  xyz_out / nrm_out: packed float[cap][3]. */
typedef struct oslam_camera {
    float fx, fy, cx, cy;          /* pixels */
    float depth_scale;             /* raw unit -> metres (0.001 for millimetre images) */
    float z_min, z_max;            /* metres */
    float max_jump;                /* metres: no normal across a larger depth step */
} oslam_camera;
int oslam_depth_to_cloud(const void *depth, int depth_is_u16, int width, int height, const oslam_camera *cam,
                         int dev, float *xyz_out, float *nrm_out, size_t cap, size_t *n_out);

/* The streaming chain in one call: depth image -> points + normals -> voxel grid (leaf > 0; 0 skips
 * it) -> scene, with the full-resolution cloud staying in HBM between the stages (only the image goes
 * up and the voxel-gridded cloud comes back for the host's reference frames).  Equivalent to
 * oslam_depth_to_cloud + oslam_voxel_grid + oslam_scene_create with the same arguments.  *n_points_out
 * (may be NULL) = points of the scene.  OSLAM_E_INVALID if fewer than 2 points remain. */
int oslam_scene_from_depth(const void *depth, int depth_is_u16, int width, int height, const oslam_camera *cam,
                           float leaf, float d_dist, unsigned ref_point_downsample_factor,
                           const oslam_params *params, oslam_scene **out, size_t *n_points_out);

/* PLY clouds with normals (host only): pcl::io::loadPLYFile<pcl::PointNormal>
 * (src/alignment.cpp:212,241) / pcl::PLYWriter (pcl/voxel_grid/voxel_grid.cpp:27-29).
 * Reads ascii and binary_little_endian; needs x y z and nx ny nz (or normal_x normal_y
 * normal_z).  *xyz_out, *nrm_out: malloc'd packed float[n][3], release with oslam_free. */
int oslam_ply_read(const char *path, float **xyz_out, float **nrm_out, size_t *n_out);
int oslam_ply_write(const char *path, const float *xyz, const float *nrm, size_t n, int binary);
void oslam_free(void *p);

/* ---- host stage (no GPU needed): accumulator peaks -> poses -> clustering.
 * Counterparts: trans_calc_kernel2, vote_weight_kernel, mat2transquat_kernel,
 * trans2idx_kernel, rot_clustering_kernel (src/cuda/kernel.cu:605-782),
 * Model::ClusterTransformations / ClusterTransformationsCPU (src/cuda/model.cu:202-266),
 * clusterPoses (src/transformation_clustering.cpp:62-137), extraction (src/cuda/ppf.cu:74-93).
 * xyz/nrm here are packed float[n][3]. */
void oslam_build_T_g(const float p[3], const float n[3], float T_rowmajor[16]);
void oslam_sort_cells(oslam_cell *cells, size_t n);                 /* count desc, code asc */
size_t oslam_filter_cells(oslam_cell *cells, size_t n, float vote_count_threshold, uint32_t max_count);
int oslam_pose_stage(const oslam_cell *cells, size_t n, const float *m_xyz, const float *m_nrm,
                     size_t M, const float *s_xyz, const float *s_nrm, size_t S, float d_dist,
                     int cpu_clustering, int use_l1_norm, int use_averaged_clusters,
                     const float *model_point_weights, float T_rowmajor[16], float *poses_out);

/* The other result fields of the reference's Model after ppf_lookup (include/model.h:100-113, read by
 * src/cuda/ppf.cu:74-93) for the cells oslam_last_cells returns, recomputed on the host from those cells
 * (the same arithmetic as the registration itself): trans_out [n][3] = transformation_trans (after the
 * clustering stage, i.e. averaged when use_averaged_clusters), rots_out [n][4] = transformation_rots
 * (w, x, y, z, kernel.cu:124-144), vote_counts_out [n] = the clustered scores, *max_idx_out = max_idx.
 * With cpu_clustering the reference fills cpu_transformations instead: vote_counts_out[0] = votes of the
 * winning cluster, its pose is the T of oslam_align.  s: the scene of that registration.  Any output may be
 * NULL; at most cap cells are written, *n_out = their number. */
int oslam_last_result(oslam_model *m, oslam_scene *s, float *trans_out, float *rots_out, float *vote_counts_out,
                      size_t cap, size_t *n_out, uint32_t *max_idx_out);
int oslam_pose_stage_ex(const oslam_cell *cells, size_t n, const float *m_xyz, const float *m_nrm,
                        size_t M, const float *s_xyz, const float *s_nrm, size_t S, float d_dist,
                        int cpu_clustering, int use_l1_norm, int use_averaged_clusters,
                        const float *model_point_weights, float T_rowmajor[16], float *poses_out,
                        float *trans_out, float *rots_out, float *scores_out, uint32_t *max_idx_out);

/* ---- multi-GPU: scene reference points shard across ranks (one process per GPU;
 * params.shard_rank / shard_world at oslam_scene_create), model tables replicated.  The reference
 * has no multi-GPU code (src/cuda/ppf.cu:45 picks one device); its one call does everything
 * (include/ppf.h:9-15), and so does oslam_align_multi.
 *
 * RCCL form.  oslam_comm wraps an RCCL communicator: rank 0 makes an id (oslam_comm_unique_id),
 * hands its OSLAM_COMM_ID_BYTES bytes to the other ranks by any means (MPI, a file, a
 * torch.distributed broadcast), every rank calls oslam_comm_create.  oslam_align_multi = this
 * rank's votes, all-reduce(MAX) of the vote maxima (the threshold is global, model.cu:164-170),
 * all-gather of the peak records above the global threshold -- device buffers end to end, exact
 * sizes, nothing truncated -- and the pose tail on the union; every rank returns the same pose. */
typedef struct oslam_comm oslam_comm;
#define OSLAM_COMM_ID_BYTES 128
int oslam_comm_unique_id(void *id_out);
int oslam_comm_create(const void *id, int rank, int world, int dev, oslam_comm **out);
void oslam_comm_destroy(oslam_comm *c);
int oslam_align_multi(oslam_model *m, oslam_scene *s, oslam_comm *c, float T_rowmajor[16], oslam_stats *stats);
/* Failure is collective: when one rank cannot go on between two collectives (no memory, too many
 * peaks, a failed kernel) an error word travels with the next collective and EVERY rank returns --
 * the failing one with its own code, the others with OSLAM_E_PEER; nobody is left waiting.  A
 * collective that fails itself aborts the communicator (ncclCommAbort): the handle then refuses
 * further calls (OSLAM_E_DEVICE) and a new one has to be made.
 *
 * Loopback communicator: `world` emulated ranks that share ONE device inside one process (one
 * thread per rank calls oslam_align_multi with its own model, scene shard and handle out[r]).  The
 * collectives become device-to-device copies between pthread barriers (which time out instead of
 * hanging).  It runs the same exchange code as RCCL does -- that is its purpose: the N > 1 state
 * machine can be executed, and its failure paths injected, on a box with one GPU. */
int oslam_comm_create_loopback(int world, int dev, oslam_comm **out /* [world] */);
int oslam_comm_info(const oslam_comm *c, int *rank, int *world, int *broken);
/* gives the communicator up without waiting for anybody (ncclCommAbort): for a rank that cannot take part in an
 * exchange its peers have entered or will enter.  The handle refuses every later call; peers that use the
 * loopback transport are released with an error, RCCL peers stay in their collective until they abort too. */
int oslam_comm_abort(oslam_comm *c);
/* test tap: the next exchange on this handle fails locally at `stage` (1 = after the votes, 2 = while
 * selecting the survivors, 3 = while growing the record buffer), as an allocation failure would */
int oslam_comm_inject_failure(oslam_comm *c, int stage);

/* A database split by MODEL instead of by reference point (SURVEY 8e's alternative; what the scenes x models
 * loop of src/cuda/ppf.cu:57-100 becomes on several GPUs when the database is large): model j of n_total lives
 * on rank j % world, `db` holds this rank's models in that order (j = rank, rank + world, ...; may be NULL on a
 * rank without models), `s` is the WHOLE scene (no reference-point shard).  Every rank registers its models
 * (oslam_db_align), then one all-gather of 17 floats per model through the communicator gives every rank all
 * poses: T_out[j*16..] for j < n_total, found_out[j] (may be NULL) = 1 when model j produced a pose, 0 when
 * nothing matched.  No exchange on the vote path. */
int oslam_db_align_multi(oslam_db *db, oslam_scene *s, oslam_comm *c, size_t n_total, float *T_out, int *found_out,
                         oslam_stats *stats_local);

/* Host-buffer form of the same exchange, for callers with their own transport (and the CPU tests
 * over gloo).  oslam_align_local runs this rank's votes and reports the number of peak records
 * above the LOCAL threshold in *n_out and the local maximum; the records stay with the model.  It
 * copies them to cells_out when they fit cap; when they do not, it copies the cap strongest and
 * returns OSLAM_E_LIMIT -- never a silent cut (cap 0 with cells_out NULL just asks for the
 * numbers and returns OSLAM_OK).  After the maxima have been exchanged, oslam_local_peaks hands
 * out the records above threshold * global_max (fewer); OSLAM_E_LIMIT with the needed number in
 * *n_out when cap is too small, call again.  Every rank, or rank 0, then calls
 * oslam_align_finish on the gathered union. */
int oslam_align_local(oslam_model *m, oslam_scene *s, oslam_cell *cells_out, size_t cap,
                      size_t *n_out, uint32_t *local_max_out, oslam_stats *stats);
int oslam_local_peaks(oslam_model *m, uint32_t global_max, oslam_cell *cells_out, size_t cap, size_t *n_out);
int oslam_align_finish(oslam_model *m, oslam_scene *s, const oslam_cell *cells, size_t n,
                       uint32_t global_max, float T_rowmajor[16], oslam_stats *stats);

/* ---- parity taps (tests): values the reference materialises as arrays.
 * Scene::getHashKeys row r (scene.cu:49-54): keys_out[n] of reference point r,
 * computed by the GPU key kernel with this d_dist (key 0 on the diagonal). */
int oslam_scene_keys(oslam_scene *s, size_t ref_index, uint32_t *keys_out);
int oslam_model_keys(oslam_model *m, size_t ref_index, uint32_t *keys_out);
/* ParallelHashArray lookup (include/impl/parallel_hash_array.hpp:80-92):
 * flat pair indices m_r*M + m_i stored under `key`, ascending; returns the
 * bucket size in *count_out and copies at most cap indices. */
int oslam_model_bucket(oslam_model *m, uint32_t key, uint32_t *pairs_out, size_t cap,
                       size_t *count_out);
/* The stored words theta_u << 11 | half << 10 | row of one key's bucket in one table slice (slices hold 2 x 1023
 * model reference points: m_r - 2046*slice = 1023*half + row), in storage order (what a vote wave streams, 4
 * consecutive words per lane). */
int oslam_model_bucket_words(oslam_model *m, uint32_t key, int slice, uint32_t *words_out, size_t cap,
                             size_t *count_out);
/* Dense accumulator acc[M][32] of scene reference point ref_index after voting. */
int oslam_vote_accumulator(oslam_model *m, oslam_scene *s, size_t ref_index, uint32_t *acc_out);
/* Cells kept by the last oslam_align / oslam_align_finish on this model, in
 * the order (count desc, code asc); poses_out (may be NULL) gets 16 floats per cell. */
int oslam_last_cells(oslam_model *m, oslam_cell *cells_out, float *poses_out, size_t cap,
                     size_t *n_out);

/* Launch stream for all kernels of this thread's calls (hipStream_t as void*;
 * NULL = the default stream).  bench.py passes torch's current stream. */
int oslam_set_stream(void *hip_stream);
/* Calls that launch on one device are serialised inside the library (one lock per device: they share
 * the device's hit-list pool, which grows to what a registration needs, at most oslam_params.scratch_gib GiB,
 * default 4, unless a single reference point needs more).  oslam_release_scratch frees the pool and the
 * other per-device work space; the next call allocates them again. */
int oslam_release_scratch(int dev);
const char *oslam_last_error(void);
/* Threads of the host stage (poses and clustering of the gathered peaks); 0 = OpenMP's default,
 * capped at 16.  Launchers that export OMP_NUM_THREADS=1 per rank can raise it here. */
int oslam_set_host_threads(int n);
/* Self-test of the device float path against the host: evaluates pm_acosf /
 * pm_atan2f / key arithmetic on `n` pseudo-random inputs on both sides and
 * returns the number of mismatching results in *mismatches. */
int oslam_selftest_math(size_t n, uint64_t seed, uint64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* OSLAM_H */
