/*
 * oslam_kernels.h -- C launch interface of the gfx950 kernels
 * (oslam_kernels.hip).  Internal to liboslam_hip.so; the public boundary is
 * include/oslam.h.  All pointers are device pointers unless noted; `stream` is
 * a hipStream_t passed as void*.  Launchers return a hipError_t as int.
 */
#ifndef OSLAM_KERNELS_H
#define OSLAM_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSLAMK_SLICE 2046      /* model reference points per table slice: two per 32-bit counter word of a row (ppf_core.h) */
#define OSLAMK_ROWS 1024       /* rows of the LDS accumulator: one per reference point of the slice + the sink row of padding entries */
#define OSLAMK_NBIN 32         /* alpha bins per accumulator row (reference uses 0..30) */

/* One slot of a slice's open-addressing table, keyed by the 32-bit PPF key. */
typedef struct oslamk_slot {
    uint32_t key;              /* 0 = empty (key 0 is never stored: kernel.cu:491) */
    uint32_t start;            /* first entry of the bucket in the entry arrays */
    uint32_t len;              /* bucket length */
    uint32_t cur;              /* fill cursor (== len after the build) */
} oslamk_slot;

/* Model pair entries, bucketed by (slice, key); every bucket starts on a multiple of 4
 * entries so that a lane can fetch 4 of them with one 16-byte load.
 *   e4[e] = theta_u << 11 | half << 10 | row   the 4 bytes a vote streams (pc_entry_word; theta_u =
 *           pc_angle_t22 of (T_m_g * m_i).y/.z, kernel.cu:330-332, in units of 2^-21 turn; half, row =
 *           pc_row11(m_r - slice*OSLAMK_SLICE)); the up to 3 padding words behind a bucket hold row
 *           1023, the accumulator's sink row
 *   uv[e] = (T_m_g * m_i).y/.z as floats: read only by the rare votes that are
 *           re-evaluated with the reference's own arithmetic (exact mode)
 *   mi[e] = m_i (parity tap only) */
typedef struct oslamk_uv {
    float uy, uz;
} oslamk_uv;

typedef struct oslamk_entries {
    uint32_t *e4;
    oslamk_uv *uv;             /* build time only (bucket order of e4); NULL afterwards and in fast mode */
    uint16_t *mi;
    /* exact mode: every bucket a second time, ordered by the position its entries' votes take inside a bin --
     * P(word) = (word * 30) mod 2^32 -- in segments of OSLAMK_PSEG entries: pw = the words, puv = their uv.  A
     * hit's votes that fall within the margin of a bin edge are the entries with (P(base) - P(word)) mod 2^32
     * below PC_T24_EDGE: a contiguous piece of this order, found by a binary search per (hit, bucket) instead of
     * a test per vote; only those (one vote in 4000) are re-evaluated with the reference's float sequence. */
    uint32_t *pw;
    oslamk_uv *puv;
    /* directory of that order: a segment of n entries at bucket offset b is cut into K = OSLAMK_PDIR_CELLS(n) cells of
     * equal width in P (one or two entries each on average); pdir[(b << OSLAMK_PDIR_SHIFT) + k], k = 0 .. K, is the position inside the
     * segment of the first entry whose cell is >= k (pdir[b + K] = n).  The search for a hit's near-edge entries is two
     * 2-byte loads and the one or two cells between them, not a binary search. */
    uint16_t *pdir;
    uint32_t n_real;           /* entries the arrays hold (the end of the last bucket, padding included): what the near-edge
                                * search checks its indices against before it loads */
} oslamk_entries;
#define OSLAMK_PSEG 4096       /* entries per sorted segment of a bucket (one LDS sort) */
/* cells of a segment of n entries: the largest power of two below the n << OSLAMK_PDIR_SHIFT directory places the
 * segment owns (so that K + 1 places fit; a bucket is padded to a multiple of 4 entries, so a segment of one still has
 * two), at most 4096: a cell must not be narrower than the margin window (2^32 / 4096 > PC_T24_EDGE) */
#ifndef OSLAMK_PDIR_SHIFT
#define OSLAMK_PDIR_SHIFT 0
#endif
#define OSLAMK_PDIR_CELLS_RAW(m) ((m) <= 1u ? 1u : 1u << (31 - __builtin_clz((m) - 1u)))
#define OSLAMK_PDIR_CELLS(n) (OSLAMK_PDIR_CELLS_RAW((n) << OSLAMK_PDIR_SHIFT) > 4096u ? 4096u : OSLAMK_PDIR_CELLS_RAW((n) << OSLAMK_PDIR_SHIFT))

typedef struct oslamk_cloud {
    const float *px, *py, *pz, *nx, *ny, *nz;
    int n;
} oslamk_cloud;

/* Bucket of one key in one slice, addressed directly by the key's slot in the union table:
 * what a vote workgroup reads instead of probing (the scene-key kernel already found the slot). */
typedef struct oslamk_uinfo {
    uint32_t start;            /* first entry of the bucket */
    uint32_t len;              /* bucket length (0 = the slice has no pair with this key); bit 31: the bucket holds a marker entry */
} oslamk_uinfo;

typedef struct oslamk_table {
    oslamk_slot *slots;        /* [n_slices][cap]: the build's counting tables (kept for the taps and the file) */
    uint32_t cap;              /* power of two */
    uint32_t shift;            /* 32 - log2(cap) */
    int n_slices;
    uint32_t *ukeys;           /* [ucap] union of the keys of all slices (0 = empty) */
    uint32_t ucap;             /* power of two */
    uint32_t ushift;
    /* reach[k1 / 32] bit k1 % 32: some key of the model can come from a pair in distance bin k1
     * (FNV collisions included), k1 < OSLAMK_REACH_BINS; pairs in other bins cannot hit */
    uint32_t *reach;
    uint32_t reach_words;      /* words of `reach` up to and including the last one that has a bit set */
    /* kmap[k1 * 17^3 + combo], k1 < kmap_bins: the number (uids) of the key that distance bin k1 and the
     * angle bins `combo` hash to (pc_key_of_bins), or OSLAMK_KMAP_NONE; covers every reachable bin unless the
     * model spans more than OSLAMK_KMAP_MAX_BINS of them (the rest are hashed and probed) */
    uint32_t *kmap;            /* holds key numbers (uids), not slots */
    uint32_t kmap_bins;
    /* uids[slot]: the dense number (0 .. n_ids-1) of the key in union-table slot `slot`.  From the scene-key
     * kernel on a key is known by this number: the hit lists sort on id_bits = ceil(log2 n_ids) bits (16 for the
     * 38 000 keys of a 5 k-point model: two 8-bit passes instead of three) and the bucket records below take
     * 8 B x n_ids per slice instead of 8 B x ucap */
    uint32_t *uids;
    uint32_t n_ids, id_bits, uinfo_stride;
    oslamk_uinfo *uinfo;       /* [n_slices][uinfo_stride], indexed by the key's number */
} oslamk_table;
#define OSLAMK_KMAP_NONE 0xffffffffu
#define OSLAMK_KMAP_MAX_BINS 2048

#define OSLAMK_REACH_BINS 16384

/* Counters one vote launch accumulates (device memory, zeroed by the host). */
typedef struct oslamk_counters {
    unsigned long long hits;
    unsigned long long votes;
    unsigned long long nonzero_cells;
    uint32_t gmax;
    uint32_t out_count;
    uint32_t redo_count;          /* vote workgroups of the current launch whose 16-bit counters overflowed (redo list length) */
    uint32_t redo_total;          /* the same, summed over the launches of a call */
    unsigned long long entries;   /* model pair entries streamed: sum of the bucket lengths over (run, slice) */
    unsigned long long items;     /* (run, slice) pairs with a bucket */
    unsigned long long prof[4];   /* -DVOTE_PROF builds: k_vote wave cycles (pre-scan, voting, wait at the barrier behind it, peak extraction) */
    uint32_t list_overflow;       /* bit 0: a hit list was too short for its hits (the counting and the hit kernel disagreed); bits 1..3: the
                                   * near-edge search met a key number, a bucket or a directory place out of range.  The call fails */
    uint32_t pad_;
} oslamk_counters;

typedef struct oslamk_cell {
    unsigned long long code;
    uint32_t count;
    uint32_t pad;
} oslamk_cell;

int oslamk_row_keys(oslamk_cloud c, int ref, float d_dist, float inv_d_dist, uint32_t *keys_out,
                    void *stream);

/* model build, pass 1: count pairs per (slice, key); n_unique[n_slices]; *overflow set if a
 * slice table filled up */
int oslamk_model_count(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                       uint32_t *n_unique, uint32_t *overflow, void *stream);
/* exclusive scan of slot.len (each rounded up to a multiple of 4) over all slots -> slot.start;
 * padded total written to *total_out */
int oslamk_table_scan(oslamk_table t, uint32_t *total_out, void *stream);
/* fill t.ukeys with every distinct key; *n_keys = number of distinct keys */
int oslamk_union_build(oslamk_table t, uint32_t *n_keys, uint32_t *overflow, void *stream);
/* fill t.reach by enumerating every key each distance bin can produce */
int oslamk_reach_build(oslamk_table t, float d_dist, void *stream);
/* t.uids[slot] = 0, 1, 2, ... for the slots of t.ukeys that hold a key; *counter (zero on entry) ends as their number */
int oslamk_union_ids(oslamk_table t, uint32_t *counter, void *stream);
/* fill t.kmap (t.kmap_bins rows) from t.ukeys / t.uids */
int oslamk_kmap_build(oslamk_table t, float d_dist, void *stream);
/* model build, pass 2: write entries. tmg = [M][8] rows y,z of T_m_g (host-computed). */
int oslamk_model_fill(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                      const float *tmg, oslamk_entries ent, void *stream);

/* model build, pass 3: inside every bucket, order the entries so that the 32 lanes the LDS serves
 * together carry evenly spaced theta_u (fewer bank conflicts of the vote atomics) */
int oslamk_bucket_spread(oslamk_table t, oslamk_entries ent, void *stream);
/* model build, pass 4 (exact mode): ent.pw / ent.puv from ent.e4 / ent.uv, see oslamk_entries */
int oslamk_bucket_psort(oslamk_table t, oslamk_entries ent, void *stream);

/* A scene pair whose key is in the model ("hit"): what a vote needs of it.  The rows y,z of
 * T_s_g * s_i (kernel.cu:334-336) that the rare exact re-evaluation needs are recomputed from the
 * scene index. */
typedef struct oslamk_pay {
    uint32_t theta_t22;        /* pc_angle_t22 of (T_s_g * s_i).y/.z */
    uint32_t idx;              /* scene index of s_i */
} oslamk_pay;

/* A run of hits of one reference point that share a key (at most 64 of them). */
typedef struct oslamk_run {
    uint32_t slot_r;           /* union-table slot of the key | (hits - 1) << OSLAMK_RUN_SHIFT */
    uint32_t first;            /* first hit of the run in the reference point's sorted hit list; bit 31: a hit of the run carries
                                * the "always re-evaluate" marker */
} oslamk_run;
#define OSLAMK_RUN_SHIFT 26    /* union tables have at most 2^26 slots */

typedef struct oslamk_vote_args {
    oslamk_cloud scene;
    const uint32_t *ref_idx;   /* [n_ref] scene indices of this shard's reference points */
    const float *tsg;          /* [n_ref][8] rows y,z of T_s_g per reference point */
    int n_ref;
    float d_dist, inv_d_dist;
    oslamk_table table;
    oslamk_entries ent;
    float thresh;              /* vote_count_threshold */
    uint32_t fixed_gmax;       /* != 0: emit cells with count > thresh*fixed_gmax only */
    oslamk_counters *counters;
    oslamk_cell *out;          /* [out_cap] */
    uint32_t out_cap;
    uint32_t *acc_dump;        /* optional [n_slices*SLICE][NBIN] for reference ordinal dump_ref */
    int dump_ref;
    int first_ref;             /* launch covers reference ordinals first_ref .. first_ref+n_launch-1 */
    int n_launch;
    int mode;                  /* 0 exact (near-edge votes re-evaluated), 1 fast (never) */
    /* Hit lists of the batch, sized by demand: reference point ref_local owns the slots
     * hit_off[ref_local] .. hit_off[ref_local + 1] of every array below; the size is the number of its
     * scene pairs whose distance bin can reach a model key (oslamk_scene_count), an upper bound of its hits.
     *   hit_key / hit_pay : hits in arrival order (oslamk_scene_hits): number of the key, payload; oslamk_sort_hits
     *                       overwrites hit_key with the key numbers in SORTED order | run-holds-a-marker << 31
     *   hit_sorted        : the payloads ordered by slot (oslamk_sort_hits) -- what oslamk_vote reads
     *   runs              : runs of equal keys in hit_sorted, run_count[ref_local] of them */
    uint32_t *keep_count;      /* [n_launch] written by oslamk_scene_count (must be zero on entry) */
    const uint32_t *hit_off;   /* [n_launch + 1] */
    uint32_t *hit_key;
    oslamk_pay *hit_pay;
    oslamk_pay *hit_sorted;
    oslamk_run *runs;
    uint32_t *hit_count;       /* [n_launch] */
    uint32_t *run_count;       /* [n_launch] */
    uint32_t *redo;            /* [vote workgroups of the batch] list of those whose 16-bit counters overflowed */
} oslamk_vote_args;

/* fill t.uinfo from the slice tables (after oslamk_table_scan / oslamk_union_build and the fill pass) */
int oslamk_uinfo_build(oslamk_table t, void *stream);

/* keep_count[ref_local] += scene pairs of the reference point whose distance bin can reach a model key */
int oslamk_scene_count(const oslamk_vote_args *a, void *stream);
/* scene pair keys -> per-reference hit lists, for reference ordinals first_ref..+n_launch-1;
 * hit_count[0..n_launch) must be zero on entry */
int oslamk_scene_hits(const oslamk_vote_args *a, void *stream);
/* hit lists -> hits_sorted (by key), same batch */
int oslamk_sort_hits(const oslamk_vote_args *a, void *stream);
/* votes of the same batch (needs the sorted hit lists) */
int oslamk_vote(const oslamk_vote_args *a, void *stream);
/* the two re-vote passes over the launch's redo list (called by oslamk_vote) */
int oslamk_vote_wide(const oslamk_vote_args *a, void *stream);
/* the votes of nm models that share the scene pass (a database group) in one grid: d_all[nm] in device memory, h_all the
 * same on the host (every member with the same batch of reference points and the same mode, its own redo list).  The
 * re-vote passes are not queued here: the caller looks at the members' redo counts and runs oslamk_vote_wide for the
 * (rare) member that has any */
int oslamk_vote_group(const oslamk_vote_args *d_all, const oslamk_vote_args *h_all, int nm, void *stream);

/* voxel grid (oslam_voxel.hip): out6 = device [n][6] (x y z nx ny nz per voxel); returns a
 * hipError_t, or -1 when the voxel count overflows int32 */
int oslamk_voxel_grid(oslamk_cloud c, float leaf, float *out6, uint32_t *n_out, void *stream);

/* depth image -> points + normals (oslam_depth.hip): d_img = device image (uint16 or float, w x h),
 * d_out6 = device [w*h][6]; the pixels that get a normal, in row-major order; returns a hipError_t */
int oslamk_depth_to_cloud(const void *d_img, int is_u16, int w, int h, float fx, float fy, float cx, float cy,
                          float scale, float z_min, float z_max, float max_jump, float *d_out6, uint32_t *n_out,
                          void *stream);

/* Device memory of the per-frame scene path (depth image, voxel grid, scene arrays).  hipFree waits for the device --
 * 0.2 ms a time, and a depth frame freed a dozen blocks -- so freed blocks are kept (per device, up to a limit) and
 * handed out again to requests they fit.  Everything on this path runs on one stream and is waited for before its
 * buffers are given back, so a block is never reused under a kernel.  oslam_dev_alloc returns a hipError_t as int;
 * oslam_release_scratch gives the kept blocks of a device back to the driver. */
int oslam_dev_alloc(void **p, size_t bytes);
void oslam_dev_free(void *p);
void oslam_dev_cache_release(int dev);

/* device [n][6] (x y z nx ny nz) -> device structure of arrays [6][n] */
int oslamk_aos6_to_soa(const float *d_in6, size_t n, float *d_soa, void *stream);

/* clustering scores of n poses (device arrays).  shash[j] = cell hash of the j-th pose in
 * (hash, pose index) order and sidx[j] its pose index; sq/st/sw = quaternions [n][4], translations [n][3], weighted
 * votes [n] in that order; cell [n][3] in pose order; score [n] in pose order.  The translation-averaging variant stays on the host. */
int oslamk_cluster_scores(int n, const int *cell, const uint32_t *shash, const uint32_t *sidx,
                          const float *sq, const float *st, const float *sw, float d_dist, int use_l1,
                          float *score, int whole_host, const unsigned long long *whole_dev, uint32_t *table, void *stream);
/* whole_host / whole_dev: the weighted votes are whole numbers with a sum below 2^24 - 1 (their float sum is then exact
 * in any order and the kernel adds them lane-parallel); whole_dev, if not NULL, points to {sum, not-whole flag} on the device */
/* table: device work space of oslamk_cluster_table_words(n) 32-bit words (the cells of the sorted list, hashed) */
size_t oslamk_cluster_table_words(int n);

/* pose tail on the device (oslam_posegpu.hip): filter, order, poses, clustering scores, winner.
 * d_Tm16 [M][16] / d_Ts16 [ceil(S/df)][16]: the frames T_g of the model points and of the scene's
 * reference-point candidates (host libm); h_rotx_cs: oslam_rotx_table.  Returns a hipError_t, -2 = no host memory */
int oslamk_pose_stage(const oslamk_cell *d_cells_in, uint32_t n_in, float min_votecount, const float *d_Tm16,
                      const float *d_Ts16, uint32_t df, const float *d_weights, const float *h_rotx_cs, float d_dist,
                      int use_l1, oslamk_cell *d_cells_out, float *d_poses, uint32_t gmax, uint32_t model_points,
                      uint32_t scene_points, int two_sorts, uint32_t *n_out, uint32_t *best_out, float T_best[16], void *stream);
/* gmax (the largest count among the records), model_points and scene_points bound the fields of a record: with few
 * enough bits the cells are ordered by one sort of packed keys (two_sorts != 0: never) */

/* The same tail in pieces, for several models in flight on one stream (a database frame): reserve once for the largest
 * record count and the number of models, enqueue every model's selection, wait, read the counts, enqueue every model's
 * chain (n >= 2 selected records), wait, read the results.  The chains share the work space: they run in stream order. */
int oslamk_pose_reserve(uint32_t n_max, uint32_t slots, const float *h_rotx_cs, void *stream);
int oslamk_pose_select_async(const oslamk_cell *d_in, uint32_t n_in, float min_votecount, oslamk_cell *d_sel, uint32_t slot,
                             void *stream);
uint32_t oslamk_pose_selected(uint32_t slot);
int oslamk_pose_finish_async(uint32_t n, const oslamk_cell *d_sel, const float *d_Tm16, const float *d_Ts16, uint32_t df,
                             const float *d_weights, float d_dist, int use_l1, oslamk_cell *d_cells_out, float *d_poses,
                             uint32_t gmax, uint32_t model_points, uint32_t scene_points, int two_sorts, uint32_t slot,
                             void *stream);
void oslamk_pose_result(uint32_t slot, uint32_t *best_out, float T_best[16]);

/* records with count > min_votecount, compacted into d_out (capacity n_in); *n_out on the host */
int oslamk_select_cells(const oslamk_cell *d_in, uint32_t n_in, float min_votecount, oslamk_cell *d_out, uint32_t *n_out,
                        void *stream);
/* frees the pose tail's work space on the current device */
void oslamk_pose_release(void);

/* device self-test: out_acos[i] = pm_acosf(x[i]); out_atan2[i] = pm_atan2f(y[i], x2[i]);
 * out_bin[i] = pc_alpha_bin_exact(...) */
int oslamk_selftest(const float *x, const float *y, const float *x2, size_t n, float *out_acos,
                    float *out_atan2, uint32_t *out_quant, uint32_t *out_bin, void *stream);

#ifdef __cplusplus
}
#endif
#endif
