/*
 * oslam_pose.c -- host stage of the PPF path (C, libm): accumulator peaks ->
 * candidate poses -> clustering -> best pose.
 *
 * Reference counterparts (pcl/alignment/): trans_calc_kernel2 and helpers
 * (src/cuda/kernel.cu:352-401,605-645), vote_weight_kernel (:766-782),
 * mat2transquat_kernel (:124-144,647-661), trans2idx_kernel (:663-699),
 * rot_clustering_kernel (:702-763), Model::ClusterTransformations and
 * ppf_lookup (src/cuda/model.cu:202-306), clusterPoses
 * (src/transformation_clustering.cpp:62-137), result extraction
 * (src/cuda/ppf.cu:74-93), ht_dist (src/cuda/linalg.cu:9-20).
 * The reference runs these as tiny GPU kernels over a few thousand poses; here
 * they run on the host after the (multi-GPU) gather of peaks.  The float
 * operation order of the reference is kept, so results do not depend on which
 * side computes them.  Known reference quirks are reproduced, not fixed:
 * kernels that return early for count <= 1 leave zero poses; the pose (0,0,0)
 * code is skipped; a pose's own votes count as 1 in clustering; the centre
 * cell is not searched; quaternions are normalised by |q|^(1/2).
 */
#include "oslam_pose.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oslam_pose_math.h"
#include "ppf_math.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define POSE_PI PM_PI_F
static int pose_threads(size_t n);
#define POSE_D PM_D_ANGLE

/* ---- 4x4 row-major helpers with the reference's evaluation order (shared with the kernels:
 * oslam_pose_math.h) ---- */
#define mat_zero pq_mat_zero
#define mat_mul pq_mat_mul

static float row_dot4(const float *row, float x, float y, float z, float w)   /* :55-57 */
{
    return row[0] * x + row[1] * y + row[2] * z + row[3] * w;
}

static void mat_translation(float x, float y, float z, float *T)   /* :170-179 */
{
    mat_zero(T);
    T[0] = 1; T[5] = 1; T[10] = 1; T[15] = 1;
    T[3] = x; T[7] = y; T[11] = z;
}

static void mat_rot(int axis, float theta, float *T)   /* rotx/roty/rotz :181-209 */
{
    float c = cosf(theta), s = sinf(theta), ms = -1 * s;
    mat_zero(T);
    T[15] = 1;
    if (axis == 0) { T[0] = 1; T[5] = c; T[9] = s; T[6] = ms; T[10] = c; }
    else if (axis == 1) { T[0] = c; T[2] = s; T[5] = 1; T[8] = ms; T[10] = c; }
    else { T[0] = c; T[4] = s; T[1] = ms; T[5] = c; T[10] = 1; }
}

#define mat_inv_rigid pq_mat_inv_rigid

void oslam_build_T_g(const float p[3], const float n[3], float T[16])
{
    float tr[16], ry[16], rz[16], tmp[16];
    float nx, ny;
    mat_translation(-1 * p[0], -1 * p[1], -1 * p[2], tr);
    mat_rot(1, atan2f(n[2], n[0]), ry);
    nx = row_dot4(ry, n[0], n[1], n[2], 1);
    ny = row_dot4(ry + 4, n[0], n[1], n[2], 1);
    mat_rot(2, -1 * atan2f(ny, nx), rz);
    mat_mul(rz, ry, tmp);
    mat_mul(tmp, tr, T);
}

/* full frames of n points (idx NULL: points 0..n-1; else point idx0 + i*step), 16 floats each */
void oslam_T_g_full(const float *xyz, const float *nrm, size_t idx0, size_t step, size_t n, float *out16)
{
    long i;
#pragma omp parallel for schedule(static) num_threads(pose_threads(n))
    for (i = 0; i < (long)n; i++) {
        const size_t r = idx0 + (size_t)i * step;
        oslam_build_T_g(xyz + 3 * r, nrm + 3 * r, out16 + 16 * (size_t)i);
    }
}

void oslam_T_g_rows(const float *xyz, const float *nrm, const uint32_t *idx, size_t n,
                    float *rows_out)
{
    long i;
#pragma omp parallel for schedule(static) num_threads(pose_threads(n))
    for (i = 0; i < (long)n; i++) {
        float T[16];
        size_t r = idx ? idx[i] : (size_t)i;
        oslam_build_T_g(xyz + 3 * r, nrm + 3 * r, T);
        memcpy(rows_out + 8 * (size_t)i, T + 4, 8 * sizeof(float));
    }
}

/* ---- cells ---- */
static int cell_order(const void *a, const void *b)
{
    const oslam_cell *x = (const oslam_cell *)a, *y = (const oslam_cell *)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return 0;
}

void oslam_sort_cells(oslam_cell *cells, size_t n) { qsort(cells, n, sizeof(oslam_cell), cell_order); }

size_t oslam_filter_cells(oslam_cell *cells, size_t n, float thresh, uint32_t gmax)
{
    float min_votecount = thresh * gmax;          /* model.cu:164 */
    size_t i, k = 0;
    for (i = 0; i < n; i++)
        if (cells[i].count > min_votecount) cells[k++] = cells[i];
    return k;
}

static int g_host_threads;      /* 0: OpenMP's default (OMP_NUM_THREADS), capped at 16 */

int oslam_set_host_threads(int n)
{
    g_host_threads = n > 0 ? n : 0;
    return OSLAM_OK;
}

/* host threads for the per-cell loops: at most 16 by default (a 1-GPU share of the host); below
 * 8192 items the loops take about a millisecond and waking threads costs more than it saves */
static int pose_threads(size_t n)
{
#ifdef _OPENMP
    if (n >= 8192) {
        if (g_host_threads) return g_host_threads;
        return omp_get_max_threads() < 16 ? omp_get_max_threads() : 16;
    }
#endif
    (void)n;
    return 1;
}

/* ---- K5: pose of one cell (kernel.cu:372-401) from the two frames ---- */
static void cell_pose(uint64_t code, const float *Tm, const float *Ts, float *T)
{
    uint32_t a = ((uint32_t)code) & 63u;
    float th = a * POSE_D - POSE_PI;
    pq_cell_pose(Tm, Ts, cosf(th), sinf(th), T);
}

/* cos and sin of the 64 rotations about x a vote code can name (the kernels' table) */
void oslam_rotx_table(float cs[128])
{
    uint32_t a;
    for (a = 0; a < 64; a++) {
        float th = a * POSE_D - POSE_PI;
        cs[2 * a] = cosf(th);
        cs[2 * a + 1] = sinf(th);
    }
}

/* T_g of every distinct point index that occurs in the cells, computed once each (many cells
 * share a reference point); `which` = 0: scene index (high word), 1: model index */
typedef struct { uint32_t *idx; float *T; size_t n; } frame_cache;

static int u32_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

static int frame_cache_build(frame_cache *fc, const oslam_cell *cells, size_t n, int which,
                             const float *xyz, const float *nrm)
{
    size_t i, u = 0;
    fc->idx = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!fc->idx) return -1;
    for (i = 0; i < n; i++)
        fc->idx[i] = which ? (((uint32_t)cells[i].code) >> 6) : (uint32_t)(cells[i].code >> 32);
    qsort(fc->idx, n, sizeof(uint32_t), u32_cmp);
    for (i = 0; i < n; i++)
        if (i == 0 || fc->idx[i] != fc->idx[i - 1]) fc->idx[u++] = fc->idx[i];
    fc->n = u;
    fc->T = (float *)malloc(sizeof(float) * 16 * (u ? u : 1));
    if (!fc->T) return -1;
    {
        long ii;
#pragma omp parallel for schedule(static) num_threads(pose_threads(u))
        for (ii = 0; ii < (long)u; ii++)
            oslam_build_T_g(xyz + 3 * (size_t)fc->idx[ii], nrm + 3 * (size_t)fc->idx[ii], fc->T + 16 * ii);
    }
    return 0;
}

static const float *frame_cache_get(const frame_cache *fc, uint32_t idx)
{
    size_t lo = 0, hi = fc->n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (fc->idx[mid] < idx) lo = mid + 1; else hi = mid;
    }
    return fc->T + 16 * lo;
}

/* ---- K7, K8: oslam_pose_math.h ---- */
#define pose_quat pq_pose_quat
#define quant_down pq_quant_down
#define fnv_cell pq_fnv_cell

typedef struct { uint32_t hash; uint32_t idx; } hash_idx;
static int hash_idx_order(const void *a, const void *b)
{
    const hash_idx *x = (const hash_idx *)a, *y = (const hash_idx *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
/* do-across loops: wait until *flag is set by the thread that owns a lower index.  Spins briefly, then yields: the
 * host may have fewer free cores than threads */
#include <sched.h>
static inline unsigned char wait_flag(const unsigned char *flag)
{
    unsigned char v;
    unsigned spins = 0;
    while (!(v = __atomic_load_n(flag, __ATOMIC_ACQUIRE)))
        if (++spins > 200) { sched_yield(); spins = 0; }
    return v;
}

static __thread oslam_cluster_hook g_cluster_hook;
void oslam_pose_set_cluster_hook(oslam_cluster_hook hook) { g_cluster_hook = hook; }

/* ---- K6 + K8 + K9 + argmax: returns max_idx ---- */
#define NO_MEMORY ((size_t)-1)
static size_t cluster_by_cells(const oslam_cell *cells, size_t n, float *trans, const float *quat,
                               float d_dist, int use_l1, int averaged, const float *weights, float *score_out)
{
    float *wv = (float *)malloc(sizeof(float) * n);
    float *score = (float *)calloc(n, sizeof(float));
    int32_t *cell = (int32_t *)malloc(sizeof(int32_t) * 3 * n);
    hash_idx *hi = (hash_idx *)malloc(sizeof(hash_idx) * n);
    const float rot_thresh = 2 * POSE_D, rot_thresh_sq = rot_thresh * rot_thresh;
    size_t i, best = 0;
    if (!wv || !score || !cell || !hi) {
        free(wv); free(score); free(cell); free(hi);
        return NO_MEMORY;
    }

    for (i = 0; i < n; i++) {
        uint32_t m = ((uint32_t)cells[i].code) >> 6;
        float w = weights ? weights[m] : 1.0f;
        int a;
        wv[i] = w * cells[i].count;                          /* kernel.cu:777 */
        for (a = 0; a < 3; a++) cell[3 * i + a] = (int32_t)(quant_down(trans[3 * i + a], d_dist) / d_dist);
        hi[i].hash = fnv_cell(cell + 3 * i);
        hi[i].idx = (uint32_t)i;
    }
    qsort(hi, n, sizeof(hash_idx), hash_idx_order);

    /* Each pose scans its 26 neighbour cells in a fixed order, so its float sums do not depend
     * on the other poses -- unless translations are averaged in place (kernel.cu:747-758): then pose i
     * sees the updated translation of every pose o < i and the original one of every o > i (the update is
     * applied in index order where the reference races).  Large sets of the plain variant go to the GPU
     * (k_cluster_scores: same sums, same order); otherwise host threads, at most 16 (a 1-GPU share of the
     * host).  The averaged variant runs on the same threads as a do-across loop: poses are handed out in
     * index order, and a pose waits for a neighbour o < i only when it meets one (the lowest unfinished
     * index never waits, so the loop cannot lock up); the reads of o > i go to a copy of the original
     * translations.  The start of a cell's poses in the sorted list comes from a table, not a search. */
    if (!averaged && n >= 2048 && g_cluster_hook &&
        g_cluster_hook(n, trans, quat, wv, cell, (const uint32_t *)hi, d_dist, use_l1, score) == 0) {
        /* scores are in place */
    } else {
        int threads = 1;
        size_t cap = 64, next = 0;
        uint32_t *rkey, *rpos;
        float *orig = NULL;
        unsigned char *done = NULL;
#ifdef _OPENMP
        threads = g_host_threads ? g_host_threads : (omp_get_max_threads() < 16 ? omp_get_max_threads() : 16);
        if (n < 512) threads = 1;
#endif
        while (cap < 2 * n) cap <<= 1;
        rkey = (uint32_t *)calloc(cap, sizeof(uint32_t));              /* cell hash -> first position in hi[]; 0 = empty */
        rpos = (uint32_t *)malloc(sizeof(uint32_t) * cap);
        if (averaged) {
            orig = (float *)malloc(sizeof(float) * 3 * n);
            done = (unsigned char *)calloc(n, 1);
        }
        if (!rkey || !rpos || (averaged && (!orig || !done))) {
            free(rkey); free(rpos); free(orig); free(done);
            free(wv); free(score); free(cell); free(hi);
            return NO_MEMORY;
        }
        if (averaged) memcpy(orig, trans, sizeof(float) * 3 * n);
        for (i = 0; i < n; i++)
            if (hi[i].hash != 0 && (i == 0 || hi[i].hash != hi[i - 1].hash)) {   /* a hash of 0 is never searched (kernel.cu:727) */
                size_t sl = (hi[i].hash * 2654435761u) & (cap - 1);
                while (rkey[sl]) sl = (sl + 1) & (cap - 1);
                rkey[sl] = hi[i].hash;
                rpos[sl] = (uint32_t)i;
            }
#pragma omp parallel num_threads(threads)
        for (;;) {
            size_t i;
            const float *q;
            float tx, ty, tz, ox, oy, oz, votes = 1;             /* kernel.cu:722 */
            int dx, dy, dz;
            i = __atomic_fetch_add(&next, 1, __ATOMIC_RELAXED);
            if (i >= n) break;
            q = quat + 4 * i;
            tx = averaged ? orig[3 * i] : trans[3 * i];
            ty = averaged ? orig[3 * i + 1] : trans[3 * i + 1];
            tz = averaged ? orig[3 * i + 2] : trans[3 * i + 2];
            ox = tx; oy = ty; oz = tz;
            for (dx = -1; dx < 2; dx++)
                for (dy = -1; dy < 2; dy++)
                    for (dz = -1; dz < 2; dz++) {
                        int32_t nb[3];
                        uint32_t h;
                        size_t j, sl;
                        if (dx == 0 && dy == 0 && dz == 0) continue;   /* kernel.cu:684-689 */
                        nb[0] = cell[3 * i] + dx; nb[1] = cell[3 * i + 1] + dy; nb[2] = cell[3 * i + 2] + dz;
                        h = fnv_cell(nb);
                        if (h == 0) continue;                          /* kernel.cu:727 */
                        sl = (h * 2654435761u) & (cap - 1);
                        while (rkey[sl] && rkey[sl] != h) sl = (sl + 1) & (cap - 1);
                        if (!rkey[sl]) continue;
                        for (j = rpos[sl]; j < n && hi[j].hash == h; j++) {
                            size_t o = hi[j].idx;
                            const float *qo = quat + 4 * o, *to;
                            float oc = wv[o];
                            float qd = fabsf(8 * (1 - (q[0] * qo[0] + q[1] * qo[1] + q[2] * qo[2] + q[3] * qo[3])));
                            if (!(qd < rot_thresh_sq)) continue;
                            to = trans + 3 * o;
                            if (averaged) {
                                if (o < i) (void)wait_flag(&done[o]);              /* its update comes first */
                                else to = orig + 3 * o;
                            }
                            if (!use_l1) {
                                float ex = tx - to[0], ey = ty - to[1], ez = tz - to[2];
                                if (!(sqrtf(ex * ex + ey * ey + ez * ez) < d_dist)) continue;
                            }
                            if (averaged) {                            /* kernel.cu:747-752 */
                                float s;
                                ox = votes * ox; oy = votes * oy; oz = votes * oz;
                                ox = ox + wv[o] * to[0];
                                oy = oy + wv[o] * to[1];
                                oz = oz + wv[o] * to[2];
                                s = 1 / (votes + oc);
                                ox = s * ox; oy = s * oy; oz = s * oz;
                            }
                            votes += oc;
                        }
                    }
            score[i] = votes;
            if (averaged) {                                            /* kernel.cu:758 */
                trans[3 * i] = ox; trans[3 * i + 1] = oy; trans[3 * i + 2] = oz;
                __atomic_store_n(&done[i], 1, __ATOMIC_RELEASE);
            }
        }
        free(rkey); free(rpos); free(orig); free(done);
    }
    for (i = 1; i < n; i++) if (score[i] > score[best]) best = i;         /* model.cu:292-295 */
    if (score_out) memcpy(score_out, score, sizeof(float) * n);           /* vote_counts_out, model.h:104 */
    free(wv); free(score); free(cell); free(hi);
    return best;
}

/* ---- Eigen-based pieces in closed form ---- */
static void rot_to_quat_xyzw(const float *R9, float q[4])   /* Eigen::Quaternionf(Matrix3f) */
{
    float t = R9[0] + R9[4] + R9[8];
    if (t > 0) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R9[7] - R9[5]) * t;
        q[1] = (R9[2] - R9[6]) * t;
        q[2] = (R9[3] - R9[1]) * t;
    } else {
        int i = 0, j, k;
        if (R9[4] > R9[0]) i = 1;
        if (R9[8] > R9[4 * i]) i = 2;
        j = (i + 1) % 3;
        k = (j + 1) % 3;
        t = sqrtf(R9[4 * i] - R9[4 * j] - R9[4 * k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (R9[3 * k + j] - R9[3 * j + k]) * t;
        q[j] = (R9[3 * j + i] + R9[3 * i + j]) * t;
        q[k] = (R9[3 * k + i] + R9[3 * i + k]) * t;
    }
}

static float relative_angle(const float *A, const float *B)   /* |angle(Ra^-1 Rb)| */
{
    float R[9], q[4], s;
    int i, j, k;
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) {
            s = 0;
            for (k = 0; k < 3; k++) s += A[4 * k + i] * B[4 * k + j];
            R[3 * i + j] = s;
        }
    rot_to_quat_xyzw(R, q);
    return fabsf(2.0f * atan2f(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), fabsf(q[3])));
}

int oslam_ht_dist(const float A[16], const float B[16], float out[2])
{
    float dx, dy, dz;
    if (!A || !B || !out) return OSLAM_E_INVALID;
    dx = A[3] - B[3]; dy = A[7] - B[7]; dz = A[11] - B[11];
    out[0] = sqrtf(dx * dx + dy * dy + dz * dz);
    out[1] = relative_angle(A, B);
    return OSLAM_OK;
}

/* greedy clustering, transformation_clustering.cpp:62-122; poses arrive sorted by votes; the first
 * cluster's average pose is the result (ppf.cu:75-77).
 *
 * The reference walks the clusters in creation order for every pose (O(poses x clusters)) and joins the
 * first whose head is within (trans_thresh, rot_thresh).  The same result here from a grid: a cluster
 * head IS a pose, a head within trans_thresh of pose p lies in one of the 27 grid cells around p (cell
 * edge a hair above trans_thresh), and "the first matching cluster" is the matching head of lowest index.
 * So pose p looks at the poses q < p of those cells, waits until each is known to be a head or a member,
 * tests the heads, and joins the lowest that matches -- or becomes a head.  Poses are handed out to the
 * host threads in index order (a do-across loop: the lowest undecided pose never waits, so it cannot lock
 * up); the votes of the clusters, the winner (first maximum in creation order) and its average are
 * serial passes in the reference's order of summation. */
typedef struct { int32_t c[3]; int32_t first; } grid_slot;

static size_t grid_find(const grid_slot *g, size_t cap, const int32_t c[3])
{
    size_t sl = ((uint32_t)c[0] * 73856093u ^ (uint32_t)c[1] * 19349663u ^ (uint32_t)c[2] * 83492791u) & (cap - 1);
    while (g[sl].first != -1 && (g[sl].c[0] != c[0] || g[sl].c[1] != c[1] || g[sl].c[2] != c[2])) sl = (sl + 1) & (cap - 1);
    return sl;
}

static int cluster_greedy(const float *T, const oslam_cell *cells, size_t n, float trans_thresh,
                          float rot_thresh, float *T_out, uint32_t *votes_out)
{
    size_t *member = (size_t *)malloc(sizeof(size_t) * n);          /* the head (a pose index) of every pose's cluster */
    uint32_t *cvotes = (uint32_t *)calloc(n, sizeof(uint32_t));       /* votes of the cluster whose head is pose p */
    int32_t *pc = (int32_t *)malloc(sizeof(int32_t) * 3 * n), *nxt = (int32_t *)malloc(sizeof(int32_t) * n);
    unsigned char *state = (unsigned char *)calloc(n, 1);             /* 0 undecided, 1 head, 2 member */
    grid_slot *grid;
    size_t cap = 64, p, win = 0, cnt = 0, next = 0;
    const double inv = trans_thresh > 0 ? 1.0 / ((double)trans_thresh * 1.0001) : 0.0;
    float ta[3] = {0, 0, 0}, qa[4] = {0, 0, 0, 0}, nq, x, y, z, w;
    int i, j, threads = 1;
#ifdef _OPENMP
    if (n >= 2048) threads = g_host_threads ? g_host_threads : (omp_get_max_threads() < 16 ? omp_get_max_threads() : 16);
#endif
    while (cap < 2 * n) cap <<= 1;
    grid = (grid_slot *)malloc(sizeof(grid_slot) * cap);
    if (!member || !cvotes || !pc || !nxt || !state || !grid || n > 0x7fffffffu) {
        free(member); free(cvotes); free(pc); free(nxt); free(state); free(grid);
        return OSLAM_E_NOMEM;
    }
    for (p = 0; p < cap; p++) grid[p].first = -1;
    /* poses into the grid, highest index first, so that every cell's list ascends */
    for (p = n; p-- > 0;) {
        const float *P = T + 16 * p;
        size_t sl;
        int a;
        for (a = 0; a < 3; a++) {
            const double v = (double)P[4 * a + 3] * inv;
            /* a translation that is not finite, or astronomically far out, matches nobody (the norm test fails):
             * any cell will do */
            pc[3 * p + a] = (v > -1e9 && v < 1e9) ? (int32_t)floor(v) : 0;
        }
        sl = grid_find(grid, cap, pc + 3 * p);
        if (grid[sl].first == -1) memcpy(grid[sl].c, pc + 3 * p, sizeof grid[sl].c);
        nxt[p] = grid[sl].first == -1 ? -2 : grid[sl].first;          /* -2 ends a list (-1 marks an empty slot) */
        grid[sl].first = (int32_t)p;
    }
#pragma omp parallel num_threads(threads)
    for (;;) {
        const size_t pp = __atomic_fetch_add(&next, 1, __ATOMIC_RELAXED);
        const float *P;
        size_t best;
        int dx, dy, dz;
        if (pp >= n) break;
        P = T + 16 * pp;
        best = pp;
        for (dx = -1; dx < 2; dx++)
            for (dy = -1; dy < 2; dy++)
                for (dz = -1; dz < 2; dz++) {
                    int32_t c[3], q;
                    size_t sl;
                    c[0] = pc[3 * pp] + dx; c[1] = pc[3 * pp + 1] + dy; c[2] = pc[3 * pp + 2] + dz;
                    sl = grid_find(grid, cap, c);
                    for (q = grid[sl].first; q >= 0 && (size_t)q < pp; q = nxt[q]) {
                        unsigned char st;
                        const float *H;
                        float ex, ey, ez;
                        if ((size_t)q > best) break;                   /* the list ascends: nothing lower than the best so far is left in it */
                        st = wait_flag(&state[q]);
                        if (st != 1) continue;                         /* only heads are compared with (transformation_clustering.cpp:78-86) */
                        H = T + 16 * (size_t)q;
                        ex = P[3] - H[3]; ey = P[7] - H[7]; ez = P[11] - H[11];
                        if (sqrtf(ex * ex + ey * ey + ez * ez) < trans_thresh && relative_angle(P, H) < rot_thresh && (size_t)q < best)
                            best = (size_t)q;
                    }
                }
        member[pp] = best;
        __atomic_store_n(&state[pp], best == pp ? 1 : 2, __ATOMIC_RELEASE);
    }
    for (p = 0; p < n; p++) cvotes[member[p]] += cells[p].count;
    for (p = 0; p < n; p++)                                            /* first of the sorted clusters: creation order = head index */
        if (state[p] == 1) { win = p; break; }
    for (p = win + 1; p < n; p++)
        if (state[p] == 1 && cvotes[p] > cvotes[win]) win = p;
    for (p = 0; p < n; p++) {
        float R[9], q[4];
        if (member[p] != win) continue;
        for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) R[3 * i + j] = T[16 * p + 4 * i + j];
        rot_to_quat_xyzw(R, q);
        ta[0] += T[16 * p + 3]; ta[1] += T[16 * p + 7]; ta[2] += T[16 * p + 11];
        for (i = 0; i < 4; i++) qa[i] += q[i];
        cnt++;
    }
    for (i = 0; i < 3; i++) ta[i] /= (float)cnt;
    for (i = 0; i < 4; i++) qa[i] /= (float)cnt;
    nq = sqrtf(qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3]);
    x = qa[0] / nq; y = qa[1] / nq; z = qa[2] / nq; w = qa[3] / nq;
    {
        float tx = 2 * x, ty = 2 * y, tz = 2 * z;
        float twx = tx * w, twy = ty * w, twz = tz * w;
        float txx = tx * x, txy = ty * x, txz = tz * x;
        float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        T_out[0] = 1 - (tyy + tzz); T_out[1] = txy - twz; T_out[2] = txz + twy; T_out[3] = ta[0];
        T_out[4] = txy + twz; T_out[5] = 1 - (txx + tzz); T_out[6] = tyz - twx; T_out[7] = ta[1];
        T_out[8] = txz - twy; T_out[9] = tyz + twx; T_out[10] = 1 - (txx + tyy); T_out[11] = ta[2];
        T_out[12] = 0; T_out[13] = 0; T_out[14] = 0; T_out[15] = 1;
    }
    if (votes_out) *votes_out = cvotes[win];
    free(member); free(cvotes); free(pc); free(nxt); free(state); free(grid);
    return OSLAM_OK;
}

static double pose_now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int oslam_pose_stage(const oslam_cell *cells, size_t n, const float *m_xyz, const float *m_nrm,
                     size_t M, const float *s_xyz, const float *s_nrm, size_t S, float d_dist,
                     int cpu_clustering, int use_l1_norm, int use_averaged_clusters,
                     const float *weights, float T_out[16], float *poses_out)
{
    return oslam_pose_stage_ex(cells, n, m_xyz, m_nrm, M, s_xyz, s_nrm, S, d_dist, cpu_clustering, use_l1_norm,
                               use_averaged_clusters, weights, T_out, poses_out, NULL, NULL, NULL, NULL);
}

/* the same with the reference's other result fields (include/model.h:100-113): trans_out [n][3] =
 * transformation_trans after the clustering stage, rots_out [n][4] = transformation_rots (w, x, y, z),
 * scores_out [n] = vote_counts_out (clustered scores; with cpu_clustering: scores_out[0] = the votes of the
 * winning cluster), *max_idx_out = max_idx.  Any of them may be NULL. */
int oslam_pose_stage_ex(const oslam_cell *cells, size_t n, const float *m_xyz, const float *m_nrm,
                        size_t M, const float *s_xyz, const float *s_nrm, size_t S, float d_dist,
                        int cpu_clustering, int use_l1_norm, int use_averaged_clusters,
                        const float *weights, float T_out[16], float *poses_out, float *trans_out, float *rots_out,
                        float *scores_out, uint32_t *max_idx_out)
{
    float *poses;
    size_t i;
    const int trace = getenv("OSLAM_TRACE_POSE") != NULL;
    double t0 = pose_now_ms(), t1, t2;
    memset(T_out, 0, 16 * sizeof(float));
    if (n == 0) return OSLAM_E_NO_VOTES;
    for (i = 0; i < n; i++) {
        uint32_t s = (uint32_t)(cells[i].code >> 32), m = ((uint32_t)cells[i].code) >> 6;
        if (s >= S || m >= M) return OSLAM_E_INVALID;
    }
    poses = (float *)calloc(16 * n, sizeof(float));
    if (!poses) return OSLAM_E_NOMEM;
    if (n > 1) {                                   /* kernel.cu:609: a single cell yields no pose */
        frame_cache fs = {0, 0, 0}, fm = {0, 0, 0};
        if (frame_cache_build(&fs, cells, n, 0, s_xyz, s_nrm) || frame_cache_build(&fm, cells, n, 1, m_xyz, m_nrm)) {
            free(fs.idx); free(fs.T); free(fm.idx); free(fm.T); free(poses);
            return OSLAM_E_NOMEM;
        }
        {
            long ii;
#pragma omp parallel for schedule(static) num_threads(pose_threads(n))
            for (ii = 0; ii < (long)n; ii++) {
                if ((cells[ii].code >> 32) == 0 && ((uint32_t)cells[ii].code) == 0) continue;   /* :628-631 */
                cell_pose(cells[ii].code, frame_cache_get(&fm, ((uint32_t)cells[ii].code) >> 6),
                          frame_cache_get(&fs, (uint32_t)(cells[ii].code >> 32)), poses + 16 * ii);
            }
        }
        free(fs.idx); free(fs.T); free(fm.idx); free(fm.T);
    }
    t1 = pose_now_ms();
    if (max_idx_out) *max_idx_out = 0;
    if (cpu_clustering) {
        uint32_t votes = 0;
        if (cluster_greedy(poses, cells, n, d_dist, POSE_D, T_out, &votes) != OSLAM_OK) {      /* model.cu:262-263 */
            free(poses);
            return OSLAM_E_NOMEM;
        }
        if (scores_out) scores_out[0] = (float)votes;
    } else if (n > 1) {
        float *trans = (float *)malloc(sizeof(float) * 3 * n);
        float *quat = (float *)malloc(sizeof(float) * 4 * n);
        size_t best;
        if (!trans || !quat) {
            free(trans); free(quat); free(poses);
            return OSLAM_E_NOMEM;
        }
        for (i = 0; i < n; i++) {
            trans[3 * i] = poses[16 * i + 3];
            trans[3 * i + 1] = poses[16 * i + 7];
            trans[3 * i + 2] = poses[16 * i + 11];
            pose_quat(poses + 16 * i, quat + 4 * i);
        }
        best = cluster_by_cells(cells, n, trans, quat, d_dist, use_l1_norm, use_averaged_clusters, weights, scores_out);
        if (best == NO_MEMORY) {
            free(trans); free(quat); free(poses);
            return OSLAM_E_NOMEM;
        }
        memcpy(T_out, poses + 16 * best, 16 * sizeof(float));
        T_out[3] = trans[3 * best]; T_out[7] = trans[3 * best + 1]; T_out[11] = trans[3 * best + 2];
        if (trans_out) memcpy(trans_out, trans, sizeof(float) * 3 * n);
        if (rots_out) memcpy(rots_out, quat, sizeof(float) * 4 * n);
        if (max_idx_out) *max_idx_out = (uint32_t)best;
        free(trans); free(quat);
    }
    t2 = pose_now_ms();
    if (trace) fprintf(stderr, "[oslam pose stage] %zu cells: poses %.2f ms, clustering %.2f ms\n", n, t1 - t0, t2 - t1);
    if (poses_out) memcpy(poses_out, poses, 16 * n * sizeof(float));
    free(poses);
    return OSLAM_OK;
}
