/*
 * oracle_ppf.c -- CPU restatement of the reference's PPF registration path.
 * TEST INFRASTRUCTURE ONLY (see oracle_ppf.h for the pinning status).
 *
 * Every function cites the reference lines it follows, relative to
 * /root/reference/pcl/alignment/.  The arithmetic is float32 with the same
 * operation order as the reference source (left-to-right sums, zero-initialised
 * 4x4 products, int/double promotions where the reference has them); build
 * with -ffp-contract=off.
 */
#include "oracle_ppf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI_F 3.141592654f                      /* CUDART_PI_F */
#define ORC_N_ANGLE 30                             /* include/kernel.h:15 */
#define ORC_D_ANGLE0 ((2.0f * (float)(ORC_PI_F)) / (float)(ORC_N_ANGLE)) /* kernel.h:16 */
#define ORC_ROT_THRESH (2 * ORC_D_ANGLE0)          /* kernel.h:17 */
#define ORC_HASH_SEED 2166136261u                  /* kernel.h:22 */

float orc_d_angle0(void) { return ORC_D_ANGLE0; }

/* src/cuda/kernel.cu:23-30 -- FNV-1a through a plain (signed) char pointer */
uint32_t orc_hash(const void *data, int n, uint32_t hash)
{
    const signed char *s = (const signed char *)data;
    while (n--) {
        hash ^= (uint32_t)(int)*s++;
        hash *= 16777619u;
    }
    return hash;
}

/* kernel.cu:51-65 */
static float dot3(orc_f3 a, orc_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static float dot4(orc_f4 a, orc_f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static float norm3(orc_f3 v) { return sqrtf(dot3(v, v)); }
static float norm4(orc_f4 v) { return sqrtf(dot4(v, v)); }

/* kernel.cu:90-92 */
static float quant_downf(float x, float y) { return x - fmodf(x, y); }

/* kernel.cu:94-100 */
orc_f4 orc_disc_feature(orc_f4 f, float d_dist, float d_angle)
{
    f.x = quant_downf(f.x, d_dist);
    f.y = quant_downf(f.y, d_angle);
    f.z = quant_downf(f.z, d_angle);
    f.w = quant_downf(f.w, d_angle);
    return f;
}

/* kernel.cu:109-122 (no clamp of the acosf argument) */
orc_f4 orc_compute_ppf(orc_f3 p1, orc_f3 n1, orc_f3 p2, orc_f3 n2)
{
    orc_f3 d;
    orc_f4 f;
    d.x = p2.x - p1.x;
    d.y = p2.y - p1.y;
    d.z = p2.z - p1.z;
    f.x = norm3(d);
    f.y = acosf(dot3(n1, d) / (norm3(n1) * norm3(d)));
    f.z = acosf(dot3(n2, d) / (norm3(n2) * norm3(d)));
    f.w = acosf(dot3(n1, n2) / (norm3(n1) * norm3(n2)));
    return f;
}

/* kernel.cu:404-457 (K1) followed by kernel.cu:460-477 (K2) for one pair */
static uint32_t pair_key(const orc_f3 *pts, const orc_f3 *nrm, int idx, int j, int df,
                         float d_dist, orc_f4 *ppf_out)
{
    orc_f4 f;
    memset(&f, 0, sizeof f);            /* thrust::device_vector zero-fills (scene.cu:70) */
    if (idx % df != 0 || j == idx) {    /* kernel.cu:432-441 */
        f.x = NAN;
        if (ppf_out) *ppf_out = f;
        return 0;                       /* kernel.cu:467-469 */
    }
    f = orc_disc_feature(orc_compute_ppf(pts[idx], nrm[idx], pts[j], nrm[j]), d_dist,
                         ORC_D_ANGLE0);
    if (ppf_out) *ppf_out = f;
    if (isnan(f.x)) return 0;
    return orc_hash(&f, (int)sizeof(orc_f4), ORC_HASH_SEED);
}

void orc_ppf_all_pairs(const orc_f3 *pts, const orc_f3 *nrm, int count, int df, float d_dist,
                       orc_f4 *ppf_out, uint32_t *keys_out)
{
    if (count <= 1) return;             /* kernel.cu:406 */
    for (int idx = 0; idx < count; idx++)
        for (int j = 0; j < count; j++)
            keys_out[(size_t)idx * count + j] =
                pair_key(pts, nrm, idx, j, df, d_dist, ppf_out ? &ppf_out[(size_t)idx * count + j] : 0);
}

void orc_ppf_row_keys(const orc_f3 *pts, const orc_f3 *nrm, int count, int ref, float d_dist,
                      uint32_t *keys_out)
{
    for (int j = 0; j < count; j++) keys_out[j] = pair_key(pts, nrm, ref, j, 1, d_dist, 0);
}

/* ---------------------------------------------------------------------------
 * include/impl/parallel_hash_array.hpp:55-77 and include/impl/util.hpp:30-52
 * -------------------------------------------------------------------------*/
typedef struct { uint32_t key; size_t idx; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);   /* radix sort is stable */
}

void orc_table_build(const uint32_t *keys, size_t n, orc_table *t)
{
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) { kv[i].key = keys[i]; kv[i].idx = i; }
    qsort(kv, n, sizeof(kv_t), kv_cmp);
    size_t nu = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || kv[i].key != kv[i - 1].key) nu++;
    t->n = n;
    t->n_unique = nu;
    t->keys = (uint32_t *)malloc(sizeof(uint32_t) * (nu ? nu : 1));
    t->counts = (size_t *)malloc(sizeof(size_t) * (nu ? nu : 1));
    t->first = (size_t *)malloc(sizeof(size_t) * (nu ? nu : 1));
    t->map = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
    size_t u = 0;
    for (size_t i = 0; i < n; i++) {
        t->map[i] = kv[i].idx;
        if (i == 0 || kv[i].key != kv[i - 1].key) {
            t->keys[u] = kv[i].key;
            t->counts[u] = 0;
            t->first[u] = i;             /* exclusive scan of counts */
            u++;
        }
        t->counts[u - 1]++;
    }
    free(kv);
}

void orc_table_free(orc_table *t)
{
    free(t->keys); free(t->counts); free(t->first); free(t->map);
    memset(t, 0, sizeof *t);
}

size_t orc_table_lower_bound(const orc_table *t, uint32_t key)
{
    size_t lo = 0, hi = t->n_unique;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (t->keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ---------------------------------------------------------------------------
 * 4x4 helpers, kernel.cu:32-49,170-299
 * -------------------------------------------------------------------------*/
static void zero4(float T[4][4]) { memset(T, 0, 16 * sizeof(float)); }

static void m_trans(orc_f3 v, float T[4][4])     /* kernel.cu:170-179 */
{
    zero4(T);
    T[0][0] = 1; T[1][1] = 1; T[2][2] = 1; T[3][3] = 1;
    T[0][3] = v.x; T[1][3] = v.y; T[2][3] = v.z;
}
static void m_rotx(float th, float T[4][4])      /* kernel.cu:181-189 */
{
    zero4(T);
    T[0][0] = 1;
    T[1][1] = cosf(th);
    T[2][1] = sinf(th);
    T[1][2] = -1 * T[2][1];
    T[2][2] = T[1][1];
    T[3][3] = 1;
}
static void m_roty(float th, float T[4][4])      /* kernel.cu:191-199 */
{
    zero4(T);
    T[0][0] = cosf(th);
    T[0][2] = sinf(th);
    T[1][1] = 1;
    T[2][0] = -1 * T[0][2];
    T[2][2] = T[0][0];
    T[3][3] = 1;
}
static void m_rotz(float th, float T[4][4])      /* kernel.cu:201-209 */
{
    zero4(T);
    T[0][0] = cosf(th);
    T[1][0] = sinf(th);
    T[0][1] = -1 * T[1][0];
    T[1][1] = T[0][0];
    T[2][2] = 1;
    T[3][3] = 1;
}
static void m_mul(const float A[4][4], const float B[4][4], float C[4][4]) /* kernel.cu:211-223 */
{
    zero4(C);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) C[i][j] += A[i][k] * B[k][j];
}
static orc_f4 m_vmul(const float A[4][4], orc_f4 b)   /* kernel.cu:234-242 */
{
    orc_f4 c, r;
    r.x = A[0][0]; r.y = A[0][1]; r.z = A[0][2]; r.w = A[0][3]; c.x = dot4(r, b);
    r.x = A[1][0]; r.y = A[1][1]; r.z = A[1][2]; r.w = A[1][3]; c.y = dot4(r, b);
    r.x = A[2][0]; r.y = A[2][1]; r.z = A[2][2]; r.w = A[2][3]; c.z = dot4(r, b);
    r.x = A[3][0]; r.y = A[3][1]; r.z = A[3][2]; r.w = A[3][3]; c.w = dot4(r, b);
    return c;
}
static orc_f4 homog(orc_f3 v) { orc_f4 w = {v.x, v.y, v.z, 1}; return w; }
static orc_f3 neg3(orc_f3 v) { orc_f3 w = {-1 * v.x, -1 * v.y, -1 * v.z}; return w; } /* vector_ops.cu:79-82 */

static void m_invht(float T[4][4], float Ti[4][4])   /* kernel.cu:254-299 */
{
    float nR[3][3];
    orc_f3 t, r0, r1, r2, tmp;
    Ti[0][0] = T[0][0]; Ti[0][1] = T[1][0]; Ti[0][2] = T[2][0];
    Ti[1][0] = T[0][1]; Ti[1][1] = T[1][1]; Ti[1][2] = T[2][1];
    Ti[2][0] = T[0][2]; Ti[2][1] = T[1][2]; Ti[2][2] = T[2][2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) nR[i][j] = -Ti[i][j];
    t.x = T[0][3]; t.y = T[1][3]; t.z = T[2][3];
    r0.x = nR[0][0]; r0.y = nR[0][1]; r0.z = nR[0][2];
    r1.x = nR[1][0]; r1.y = nR[1][1]; r1.z = nR[1][2];
    r2.x = nR[2][0]; r2.y = nR[2][1]; r2.z = nR[2][2];
    tmp.x = dot3(r0, t); tmp.y = dot3(r1, t); tmp.z = dot3(r2, t);   /* mat3f_vmul :225-232 */
    Ti[0][3] = tmp.x; Ti[1][3] = tmp.y; Ti[2][3] = tmp.z;
    Ti[3][0] = 0; Ti[3][1] = 0; Ti[3][2] = 0; Ti[3][3] = 1;
}

/* T_g = R_z * R_y * Trans(-p): kernel.cu:310-318 (== :352-369 + :382-388) */
static void build_T_g(orc_f3 p, orc_f3 n, float T_g[4][4])
{
    float transm[4][4], rot_y[4][4], rot_z[4][4], T_tmp[4][4];
    orc_f4 n_tmp;
    m_trans(neg3(p), transm);
    m_roty(atan2f(n.z, n.x), rot_y);
    n_tmp = m_vmul(rot_y, homog(n));
    m_rotz(-1 * atan2f(n_tmp.y, n_tmp.x), rot_z);
    m_mul(rot_z, rot_y, T_tmp);
    m_mul(T_tmp, transm, T_g);
}

/* kernel.cu:338-342 given u = T_m_g*m_i and v = T_s_g*s_i */
static unsigned alpha_idx_from_uv(orc_f3 u, orc_f3 v, float *alpha_q)
{
    u.x = 0;
    v.x = 0;
    float cx = u.y * v.z - u.z * v.y;           /* cross(u,v).x, kernel.cu:84 */
    float alpha = atan2f(cx, dot3(u, v));
    alpha = quant_downf(alpha + ORC_PI_F, ORC_D_ANGLE0);
    if (alpha_q) *alpha_q = alpha;
    return (unsigned)lrintf(alpha / ORC_D_ANGLE0);
}

/* kernel.cu:302-349 */
unsigned orc_trans_model_scene(orc_f3 m_r, orc_f3 n_r_m, orc_f3 m_i, orc_f3 s_r, orc_f3 n_r_s,
                               orc_f3 s_i, float T_out[16])
{
    float T_m_g[4][4], T_s_g[4][4], rot_x[4][4], T_tmp[4][4], T_tmp2[4][4], T[4][4];
    orc_f4 n_tmp;
    orc_f3 u, v;
    float alpha;
    build_T_g(m_r, n_r_m, T_m_g);
    build_T_g(s_r, n_r_s, T_s_g);
    n_tmp = m_vmul(T_m_g, homog(m_i)); u.x = n_tmp.x; u.y = n_tmp.y; u.z = n_tmp.z;
    n_tmp = m_vmul(T_s_g, homog(s_i)); v.x = n_tmp.x; v.y = n_tmp.y; v.z = n_tmp.z;
    unsigned idx = alpha_idx_from_uv(u, v, &alpha);
    if (T_out) {
        m_rotx(alpha, rot_x);
        m_invht(T_s_g, T_tmp);
        m_mul(T_tmp, rot_x, T_tmp2);
        m_mul(T_tmp2, T_m_g, T);
        memcpy(T_out, T, sizeof T);
    }
    return idx;
}

/* ---------------------------------------------------------------------------
 * Votes, literal: model.cu:43-82 (model table) + model.cu:95-171
 * -------------------------------------------------------------------------*/
static int u64_cmp(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}
/* order used everywhere thrust leaves ties unspecified: count desc, code asc */
static int cell_cmp(const void *a, const void *b)
{
    const orc_cell *x = (const orc_cell *)a, *y = (const orc_cell *)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    return x->code < y->code ? -1 : (x->code > y->code);
}

static orc_cell *threshold_cells(orc_cell *cells, size_t n, float thresh, size_t *n_out,
                                 orc_stats *st)
{
    qsort(cells, n, sizeof(orc_cell), cell_cmp);              /* model.cu:155-158 */
    if (n == 0) { *n_out = 0; if (st) { st->max_count = 0; st->num_top = 0; } return cells; }
    float min_votecount = thresh * cells[0].count;            /* model.cu:164 */
    size_t top = 0;
    for (size_t i = 0; i < n; i++) if (cells[i].count > min_votecount) top++;  /* :165-167 */
    if (st) { st->max_count = cells[0].count; st->num_top = top; }
    *n_out = top;                                             /* resize :169-170 */
    return cells;
}

orc_cell *orc_votes_literal(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                            const orc_f3 *sn, int S, int df, float d_dist, float thresh,
                            size_t *n_out, orc_stats *st, orc_cell **all_cells, size_t *all_n)
{
    orc_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof *st);
    size_t MM = (size_t)M * M, SS = (size_t)S * S;
    uint32_t *mkeys = (uint32_t *)calloc(MM ? MM : 1, sizeof(uint32_t));
    uint32_t *skeys = (uint32_t *)calloc(SS ? SS : 1, sizeof(uint32_t));
    orc_ppf_all_pairs(mp, mn, M, 1, d_dist, 0, mkeys);        /* Model ctor, df = 1 */
    orc_ppf_all_pairs(sp, sn, S, df, d_dist, 0, skeys);       /* Scene ctor */
    orc_table tab;
    orc_table_build(mkeys, MM, &tab);                         /* model.cu:79 */
    st->num_model_keys = tab.n_unique;
    free(mkeys);

    /* K3 + scan: model.cu:96-121 */
    size_t *offs = (size_t *)malloc(sizeof(size_t) * (SS + 1));
    size_t V = 0;
    for (size_t idx = 0; idx < SS; idx++) {
        uint32_t k = skeys[idx];
        size_t cnt = 0;
        if ((idx / S) % df == 0 && (idx % S) != (idx / S)) st->num_scene_ppfs++;
        if (k != 0) {
            size_t ti = orc_table_lower_bound(&tab, k);
            if (ti < tab.n_unique && tab.keys[ti] == k) { cnt = tab.counts[ti]; st->num_hits++; }
        }
        offs[idx] = V;
        V += cnt;
    }
    offs[SS] = V;
    st->num_votes = V;

    /* K4: kernel.cu:504-554 */
    uint64_t *votes = (uint64_t *)malloc(sizeof(uint64_t) * (V ? V : 1));
    for (size_t idx = 0; idx < SS; idx++) {
        if (offs[idx + 1] == offs[idx]) continue;
        size_t ti = orc_table_lower_bound(&tab, skeys[idx]);
        unsigned s_r = (unsigned)(idx / S), s_i = (unsigned)(idx - (size_t)s_r * S);
        for (size_t i = 0; i < tab.counts[ti]; i++) {
            size_t mp_idx = tab.map[tab.first[ti] + i];
            unsigned m_r = (unsigned)(mp_idx / M), m_i = (unsigned)(mp_idx - (size_t)m_r * M);
            unsigned a = orc_trans_model_scene(mp[m_r], mn[m_r], mp[m_i], sp[s_r], sn[s_r], sp[s_i], 0);
            votes[offs[idx] + i] = (((uint64_t)s_r) << 32) | (uint32_t)(m_r << 6) | a;
        }
    }
    free(offs);
    free(skeys);
    orc_table_free(&tab);

    /* sort + histogram: model.cu:148-152 */
    qsort(votes, V, sizeof(uint64_t), u64_cmp);
    size_t nu = 0;
    for (size_t i = 0; i < V; i++) if (i == 0 || votes[i] != votes[i - 1]) nu++;
    st->num_unique_votes = nu;
    orc_cell *cells = (orc_cell *)malloc(sizeof(orc_cell) * (nu ? nu : 1));
    size_t u = 0;
    for (size_t i = 0; i < V; i++) {
        if (i == 0 || votes[i] != votes[i - 1]) { cells[u].code = votes[i]; cells[u].count = 0; u++; }
        cells[u - 1].count++;
    }
    free(votes);
    if (all_cells) {
        *all_cells = (orc_cell *)malloc(sizeof(orc_cell) * (nu ? nu : 1));
        memcpy(*all_cells, cells, sizeof(orc_cell) * nu);
        *all_n = nu;
    }
    return threshold_cells(cells, nu, thresh, n_out, st);
}

/* ---------------------------------------------------------------------------
 * Votes, fused: same cells, dense per-reference accumulator.
 * -------------------------------------------------------------------------*/
typedef struct {
    orc_table tab;
    float *T_m_g;     /* [M][16] */
    float *ent_uy;    /* [M*M] in table (map) order: (T_m_g * m_i).y */
    float *ent_uz;
    uint32_t *ent_mr;
    int M;
} fused_model;

static void fused_model_build(fused_model *fm, const orc_f3 *mp, const orc_f3 *mn, int M,
                              float d_dist)
{
    size_t MM = (size_t)M * M;
    uint32_t *mkeys = (uint32_t *)calloc(MM ? MM : 1, sizeof(uint32_t));
    orc_ppf_all_pairs(mp, mn, M, 1, d_dist, 0, mkeys);
    orc_table_build(mkeys, MM, &fm->tab);
    free(mkeys);
    fm->M = M;
    fm->T_m_g = (float *)malloc(sizeof(float) * 16 * (M ? M : 1));
    for (int r = 0; r < M; r++) build_T_g(mp[r], mn[r], (float(*)[4])(fm->T_m_g + 16 * r));
    fm->ent_uy = (float *)malloc(sizeof(float) * (MM ? MM : 1));
    fm->ent_uz = (float *)malloc(sizeof(float) * (MM ? MM : 1));
    fm->ent_mr = (uint32_t *)malloc(sizeof(uint32_t) * (MM ? MM : 1));
    for (size_t e = 0; e < MM; e++) {
        size_t flat = fm->tab.map[e];
        unsigned m_r = (unsigned)(flat / M), m_i = (unsigned)(flat - (size_t)m_r * M);
        orc_f4 u = m_vmul((const float(*)[4])(fm->T_m_g + 16 * m_r), homog(mp[m_i]));
        fm->ent_mr[e] = m_r;
        fm->ent_uy[e] = u.y;
        fm->ent_uz[e] = u.z;
    }
}
static void fused_model_free(fused_model *fm)
{
    orc_table_free(&fm->tab);
    free(fm->T_m_g); free(fm->ent_uy); free(fm->ent_uz); free(fm->ent_mr);
}

/* accumulate all votes of scene reference point s_r into acc[M][32] */
static void fused_accumulate(const fused_model *fm, const orc_f3 *sp, const orc_f3 *sn, int S,
                             int s_r, float d_dist, uint32_t *acc, uint64_t *hits, uint64_t *votes)
{
    float T_s_g[4][4];
    build_T_g(sp[s_r], sn[s_r], T_s_g);
    for (int i = 0; i < S; i++) {
        uint32_t k = pair_key(sp, sn, s_r, i, 1, d_dist, 0);
        if (k == 0) continue;                                   /* kernel.cu:491,520 */
        size_t ti = orc_table_lower_bound(&fm->tab, k);
        if (ti >= fm->tab.n_unique || fm->tab.keys[ti] != k) continue;
        orc_f4 v4 = m_vmul(T_s_g, homog(sp[i]));
        orc_f3 v = {v4.x, v4.y, v4.z};
        size_t first = fm->tab.first[ti], cnt = fm->tab.counts[ti];
        (*hits)++;
        (*votes) += cnt;
        for (size_t e = first; e < first + cnt; e++) {
            orc_f3 u = {0, fm->ent_uy[e], fm->ent_uz[e]};
            unsigned a = alpha_idx_from_uv(u, v, 0);
            acc[(size_t)fm->ent_mr[e] * 32 + a]++;
        }
    }
}

void orc_accumulator_for_ref(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                             const orc_f3 *sn, int S, int s_r, float d_dist, uint32_t *acc)
{
    fused_model fm;
    uint64_t h = 0, v = 0;
    fused_model_build(&fm, mp, mn, M, d_dist);
    memset(acc, 0, sizeof(uint32_t) * 32 * (size_t)M);
    fused_accumulate(&fm, sp, sn, S, s_r, d_dist, acc, &h, &v);
    fused_model_free(&fm);
}

void *orc_fused_create(const orc_f3 *mp, const orc_f3 *mn, int M, float d_dist)
{
    fused_model *fm = (fused_model *)calloc(1, sizeof *fm);
    fused_model_build(fm, mp, mn, M, d_dist);
    return fm;
}

void orc_fused_free(void *h)
{
    if (!h) return;
    fused_model_free((fused_model *)h);
    free(h);
}

orc_cell *orc_fused_votes(void *h, const orc_f3 *sp, const orc_f3 *sn, int S, int df, float d_dist,
                          float thresh, long ref_begin, long ref_step, long ref_limit, int threads,
                          size_t *n_out, orc_stats *st)
{
    const fused_model *fmp = (const fused_model *)h;
    const int M = fmp->M;
    orc_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof *st);
    st->num_model_keys = fmp->tab.n_unique;

    long n_ref_all = (S + df - 1) / df;
    if (ref_step < 1) ref_step = 1;
    long n_ref = 0;
    for (long k = ref_begin; k < n_ref_all; k += ref_step) n_ref++;
    if (ref_limit >= 0 && n_ref > ref_limit) n_ref = ref_limit;

    size_t cap = 1 << 16, ncell = 0;
    orc_cell *cells = (orc_cell *)malloc(sizeof(orc_cell) * cap);
    uint32_t gmax = 0;
    uint64_t hits = 0, votes = 0, uniq = 0, ppfs = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel
    {
        uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * 32 * (size_t)(M ? M : 1));
#pragma omp for schedule(dynamic, 1) reduction(+ : hits, votes, uniq, ppfs)
        for (long kk = 0; kk < n_ref; kk++) {
            int s_r = (int)(df * (ref_begin + kk * ref_step));
            memset(acc, 0, sizeof(uint32_t) * 32 * (size_t)M);
            fused_accumulate(fmp, sp, sn, S, s_r, d_dist, acc, &hits, &votes);
            ppfs += (uint64_t)(S - 1);
            uint32_t lmax = 0;
            for (size_t c = 0; c < (size_t)M * 32; c++) {
                if (acc[c]) uniq++;
                if (acc[c] > lmax) lmax = acc[c];
            }
            uint32_t g;
#pragma omp critical(orc_gmax)
            {
                if (lmax > gmax) gmax = lmax;
                g = gmax;
            }
            float bound = thresh * g;        /* <= final threshold: emits a superset */
            for (size_t c = 0; c < (size_t)M * 32; c++) {
                if (acc[c] > bound) {
#pragma omp critical(orc_cells)
                    {
                        if (ncell == cap) { cap *= 2; cells = (orc_cell *)realloc(cells, sizeof(orc_cell) * cap); }
                        cells[ncell].code = (((uint64_t)(unsigned)s_r) << 32) |
                                            (uint32_t)(((unsigned)(c / 32)) << 6) | (unsigned)(c % 32);
                        cells[ncell].count = acc[c];
                        ncell++;
                    }
                }
            }
        }
        free(acc);
    }
    st->num_scene_ppfs = ppfs;
    st->num_hits = hits;
    st->num_votes = votes;
    st->num_unique_votes = uniq;
    return threshold_cells(cells, ncell, thresh, n_out, st);
}

orc_cell *orc_votes_fused(const orc_f3 *mp, const orc_f3 *mn, int M, const orc_f3 *sp,
                          const orc_f3 *sn, int S, int df, float d_dist, float thresh,
                          long ref_begin, long ref_step, long ref_limit, int threads,
                          size_t *n_out, orc_stats *st)
{
    void *h = orc_fused_create(mp, mn, M, d_dist);
    orc_cell *cells = orc_fused_votes(h, sp, sn, S, df, d_dist, thresh, ref_begin, ref_step, ref_limit,
                                      threads, n_out, st);
    orc_fused_free(h);
    return cells;
}

/* ---------------------------------------------------------------------------
 * Poses: K5 kernel.cu:352-401,605-645
 * -------------------------------------------------------------------------*/
void orc_trans_calc2(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                     const orc_f3 *sp, const orc_f3 *sn, float *T_out)
{
    memset(T_out, 0, sizeof(float) * 16 * n);     /* device_vector zero-fill, model.cu:192 */
    if (n <= 1) return;                           /* kernel.cu:609 */
    for (size_t idx = 0; idx < n; idx++) {
        unsigned s = (unsigned)(cells[idx].code >> 32);
        unsigned mac = (unsigned)(cells[idx].code & 0xffffffffu);
        unsigned m = mac >> 6, a = mac & 63u;
        if (s == 0 && m == 0 && a == 0) continue; /* kernel.cu:628-631 */
        float T_m_g[4][4], T_s_g[4][4], rot_x[4][4], T_tmp[4][4], T_tmp2[4][4], T[4][4];
        build_T_g(mp[m], mn[m], T_m_g);
        build_T_g(sp[s], sn[s], T_s_g);
        m_rotx(a * ORC_D_ANGLE0 - ORC_PI_F, rot_x);   /* kernel.cu:397: lower bin edge */
        m_invht(T_s_g, T_tmp);
        m_mul(T_tmp, rot_x, T_tmp2);
        m_mul(T_tmp2, T_m_g, T);
        memcpy(T_out + 16 * idx, T, sizeof T);
    }
}

/* kernel.cu:128-144 -- q = (w,x,y,z) stored in (.x,.y,.z,.w); note the
 * normalisation by sqrt(norm(q)) = |q|^(1/2).  sqrt(float) resolves to the
 * float overload in CUDA C++; the 0.5* factors are double but exact. */
static orc_f4 hrotmat2quat(const float *Tp)
{
    const float(*T)[4] = (const float(*)[4])Tp;
    float t, r;
    orc_f4 q;
    t = T[0][0] + T[1][1] + T[2][2];
    r = sqrtf(1 + t);
    q.x = (float)(0.5 * r);
    q.y = copysignf((float)(0.5 * sqrtf(1 + T[0][0] - T[1][1] - T[2][2])), T[2][1] - T[1][2]);
    q.z = copysignf((float)(0.5 * sqrtf(1 - T[0][0] + T[1][1] - T[2][2])), T[0][2] - T[2][0]);
    q.w = copysignf((float)(0.5 * sqrtf(1 - T[0][0] - T[1][1] + T[2][2])), T[1][0] - T[0][1]);
    float n = sqrtf(norm4(q));
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
    return q;
}

/* K7: kernel.cu:647-661 */
void orc_mat2transquat(const float *T, size_t n, orc_f3 *trans, orc_f4 *quat)
{
    memset(trans, 0, sizeof(orc_f3) * n);
    memset(quat, 0, sizeof(orc_f4) * n);
    if (n <= 1) return;                           /* kernel.cu:651 */
    for (size_t i = 0; i < n; i++) {
        trans[i].x = T[16 * i + 3]; trans[i].y = T[16 * i + 7]; trans[i].z = T[16 * i + 11];
        quat[i] = hrotmat2quat(T + 16 * i);
    }
}

/* K8: kernel.cu:102-107,663-699 */
void orc_trans2idx(const orc_f3 *trans, size_t n, float d_dist, uint32_t *trans_hash,
                   uint32_t *adjacent)
{
    memset(trans_hash, 0, sizeof(uint32_t) * n);
    memset(adjacent, 0, sizeof(uint32_t) * 27 * n);
    if (n <= 1) return;                           /* kernel.cu:667 */
    for (size_t idx = 0; idx < n; idx++) {
        int cell[3], adj[3];
        cell[0] = (int)(quant_downf(trans[idx].x, d_dist) / d_dist);
        cell[1] = (int)(quant_downf(trans[idx].y, d_dist) / d_dist);
        cell[2] = (int)(quant_downf(trans[idx].z, d_dist) / d_dist);
        trans_hash[idx] = orc_hash(cell, 12, ORC_HASH_SEED);
        int c = 0;
        for (int i = -1; i < 2; i++)
            for (int j = -1; j < 2; j++)
                for (int k = -1; k < 2; k++, c++) {
                    if (i == 0 && j == 0 && k == 0) { adjacent[27 * idx + c] = 0; continue; }
                    adj[0] = cell[0] + i; adj[1] = cell[1] + j; adj[2] = cell[2] + k;
                    adjacent[27 * idx + c] = orc_hash(adj, 12, ORC_HASH_SEED);
                }
    }
}

/* model.cu:173-189 (K6 = vote_weight_kernel, kernel.cu:766-782: weight of the cell's model point times
 * its count; the weights are all 1.0 unless SetModelPointVoteWeights was called, model.cu:67,84-93),
 * :202-244 (K7 done by caller, K8, table, lower_bound, K9 kernel.cu:702-763), :292-295 (argmax).
 * model_point_weights == NULL: all 1.0. */
size_t orc_cluster_gpu_style_w(const orc_cell *cells, size_t n, orc_f3 *trans, const orc_f4 *quat,
                               float d_dist, int use_l1_norm, int use_averaged_clusters,
                               const float *model_point_weights, float *scores_out)
{
    float *weighted = (float *)malloc(sizeof(float) * (n ? n : 1));
    if (n > 1)
        for (size_t i = 0; i < n; i++) {                                           /* K6 */
            uint32_t model_point_idx = ((uint32_t)cells[i].code) >> 6;            /* kernel.cu:774-775 */
            weighted[i] = (model_point_weights ? model_point_weights[model_point_idx] : 1.0f) * cells[i].count;
        }
    else memset(weighted, 0, sizeof(float) * n);                                   /* kernel.cu:769 */
    uint32_t *th = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *adj = (uint32_t *)malloc(sizeof(uint32_t) * 27 * (n ? n : 1));
    orc_trans2idx(trans, n, d_dist, th, adj);
    orc_table tab;
    orc_table_build(th, n, &tab);
    memset(scores_out, 0, sizeof(float) * n);
    if (n > 1) {                                  /* kernel.cu:712 */
        float rot_thresh_sq = ORC_ROT_THRESH * ORC_ROT_THRESH;
        for (size_t idx = 0; idx < n; idx++) {
            orc_f3 thisTrans = trans[idx];
            orc_f4 thisQuat = quat[idx];
            float vote_count_out = 1;             /* kernel.cu:722 */
            orc_f3 out = thisTrans;
            for (int b = 0; b < 27; b++) {
                uint32_t h = adj[27 * idx + b];
                if (h == 0) continue;
                size_t ti = orc_table_lower_bound(&tab, h);
                if (ti >= tab.n_unique || tab.keys[ti] != h) continue;
                for (size_t j = 0; j < tab.counts[ti]; j++) {
                    size_t o = tab.map[tab.first[ti] + j];
                    float thisVoteCount = weighted[o];
                    float quatDiff = fabsf(8 * (1 - dot4(thisQuat, quat[o])));
                    if (quatDiff < rot_thresh_sq) {
                        if (!use_l1_norm) {
                            orc_f3 d = {thisTrans.x - trans[o].x, thisTrans.y - trans[o].y,
                                        thisTrans.z - trans[o].z};
                            if (!(norm3(d) < d_dist)) continue;
                        }
                        if (use_averaged_clusters) {
                            out.x = vote_count_out * out.x; out.y = vote_count_out * out.y; out.z = vote_count_out * out.z;
                            out.x = out.x + weighted[o] * trans[o].x;
                            out.y = out.y + weighted[o] * trans[o].y;
                            out.z = out.z + weighted[o] * trans[o].z;
                            float s = 1 / (vote_count_out + thisVoteCount);
                            out.x = s * out.x; out.y = s * out.y; out.z = s * out.z;
                        }
                        vote_count_out += thisVoteCount;
                    }
                }
            }
            scores_out[idx] = vote_count_out;
            trans[idx] = out;                     /* in place, kernel.cu:758 (serial order) */
        }
    }
    size_t best = 0;                              /* thrust::max_element: first maximum */
    for (size_t i = 1; i < n; i++) if (scores_out[i] > scores_out[best]) best = i;
    orc_table_free(&tab);
    free(weighted); free(th); free(adj);
    return best;
}

size_t orc_cluster_gpu_style(const orc_cell *cells, size_t n, orc_f3 *trans, const orc_f4 *quat,
                             float d_dist, int use_l1_norm, int use_averaged_clusters,
                             float *scores_out)
{
    return orc_cluster_gpu_style_w(cells, n, trans, quat, d_dist, use_l1_norm, use_averaged_clusters, 0, scores_out);
}

/* ---------------------------------------------------------------------------
 * Eigen-dependent parts, restated in closed form (parity unpinned).
 * -------------------------------------------------------------------------*/
/* Eigen::Quaternionf(Matrix3f): returns (x,y,z,w) */
static void quat_from_R(const float R[3][3], float q[4])
{
    float t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R[2][1] - R[1][2]) * t;
        q[1] = (R[0][2] - R[2][0]) * t;
        q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(R[i][i] - R[j][j] - R[k][k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (R[k][j] - R[j][k]) * t;
        q[j] = (R[j][i] + R[i][j]) * t;
        q[k] = (R[k][i] + R[i][k]) * t;
    }
}
/* |AngleAxisf(Ra^-1 * Rb).angle()| */
static float rel_angle(const float *A, const float *B)
{
    float R[3][3], q[4];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float s = 0;
            for (int k = 0; k < 3; k++) s += A[4 * k + i] * B[4 * k + j];
            R[i][j] = s;
        }
    quat_from_R(R, q);
    float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    return fabsf(2.0f * atan2f(nrm, fabsf(q[3])));
}

/* src/cuda/linalg.cu:9-20 */
void orc_ht_dist(const float A[16], const float B[16], float out[2])
{
    orc_f3 d = {A[3] - B[3], A[7] - B[7], A[11] - B[11]};
    out[0] = norm3(d);
    out[1] = rel_angle(A, B);
}

/* src/transformation_clustering.cpp:127-137 */
static int poses_within(const float *P1, const float *P2, float trans_thresh, float rot_thresh)
{
    orc_f3 d = {P1[3] - P2[3], P1[7] - P2[7], P1[11] - P2[11]};
    return norm3(d) < trans_thresh && rel_angle(P1, P2) < rot_thresh;
}

/* src/transformation_clustering.cpp:62-122; input poses are already in
 * (votes desc, code asc) order (std::sort there is unstable: tie order is ours) */
int orc_cluster_poses_cpu(const float *T, const orc_cell *cells, size_t n, float trans_thresh,
                          float rot_thresh, float *T_out, uint32_t *votes_out)
{
    size_t *cl_first = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));   /* front() pose */
    uint32_t *cl_votes = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    size_t *member = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));     /* cluster of pose */
    size_t ncl = 0;
    for (size_t p = 0; p < n; p++) {
        int found = 0;
        for (size_t c = 0; c < ncl; c++) {
            if (poses_within(T + 16 * p, T + 16 * cl_first[c], trans_thresh, rot_thresh)) {
                found = 1; member[p] = c; cl_votes[c] += cells[p].count; break;
            }
        }
        if (!found) { cl_first[ncl] = p; cl_votes[ncl] = cells[p].count; member[p] = ncl; ncl++; }
    }
    /* sort clusters by votes desc (ties: creation order) */
    size_t *order = (size_t *)malloc(sizeof(size_t) * (ncl ? ncl : 1));
    for (size_t c = 0; c < ncl; c++) order[c] = c;
    for (size_t a = 1; a < ncl; a++) {
        size_t v = order[a], b = a;
        while (b > 0 && cl_votes[order[b - 1]] < cl_votes[v]) { order[b] = order[b - 1]; b--; }
        order[b] = v;
    }
    int nres = ncl < 3 ? (int)ncl : 3;
    for (int r = 0; r < nres; r++) {
        size_t c = order[r];
        float ta[3] = {0, 0, 0}, qa[4] = {0, 0, 0, 0};
        size_t cnt = 0;
        for (size_t p = 0; p < n; p++) {
            if (member[p] != c) continue;
            float R[3][3], q[4];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = T[16 * p + 4 * i + j];
            quat_from_R(R, q);
            ta[0] += T[16 * p + 3]; ta[1] += T[16 * p + 7]; ta[2] += T[16 * p + 11];
            for (int i = 0; i < 4; i++) qa[i] += q[i];
            cnt++;
        }
        for (int i = 0; i < 3; i++) ta[i] /= (float)cnt;
        for (int i = 0; i < 4; i++) qa[i] /= (float)cnt;
        float nq = sqrtf(qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3]);
        float x = qa[0] / nq, y = qa[1] / nq, z = qa[2] / nq, w = qa[3] / nq;
        float *O = T_out + 16 * r;
        /* Eigen::Quaternion::toRotationMatrix */
        float tx = 2 * x, ty = 2 * y, tz = 2 * z;
        float twx = tx * w, twy = ty * w, twz = tz * w;
        float txx = tx * x, txy = ty * x, txz = tz * x;
        float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        O[0] = 1 - (tyy + tzz); O[1] = txy - twz;       O[2] = txz + twy;       O[3] = ta[0];
        O[4] = txy + twz;       O[5] = 1 - (txx + tzz); O[6] = tyz - twx;       O[7] = ta[1];
        O[8] = txz - twy;       O[9] = tyz + twx;       O[10] = 1 - (txx + tyy); O[11] = ta[2];
        O[12] = 0; O[13] = 0; O[14] = 0; O[15] = 1;
        if (votes_out) votes_out[r] = cl_votes[c];
    }
    free(cl_first); free(cl_votes); free(member); free(order);
    return nres;
}

/* model.cu:269-306 + ppf.cu:74-93 */
int orc_pose_from_cells_w(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                          const orc_f3 *sp, const orc_f3 *sn, float d_dist, int cpu_clustering,
                          int use_l1_norm, int use_averaged_clusters, const float *model_point_weights,
                          float T_out[16])
{
    memset(T_out, 0, 16 * sizeof(float));
    if (n == 0) return 1;
    float *T = (float *)malloc(sizeof(float) * 16 * n);
    orc_trans_calc2(cells, n, mp, mn, sp, sn, T);
    if (cpu_clustering) {
        float res[48];
        int k = orc_cluster_poses_cpu(T, cells, n, d_dist, ORC_D_ANGLE0, res, 0);
        if (k > 0) memcpy(T_out, res, 16 * sizeof(float));
    } else {
        orc_f3 *tr = (orc_f3 *)malloc(sizeof(orc_f3) * n);
        orc_f4 *qu = (orc_f4 *)malloc(sizeof(orc_f4) * n);
        float *sc = (float *)malloc(sizeof(float) * n);
        orc_mat2transquat(T, n, tr, qu);
        size_t best = orc_cluster_gpu_style_w(cells, n, tr, qu, d_dist, use_l1_norm,
                                              use_averaged_clusters, model_point_weights, sc);
        memcpy(T_out, T + 16 * best, 16 * sizeof(float));
        T_out[3] = tr[best].x; T_out[7] = tr[best].y; T_out[11] = tr[best].z;  /* ppf.cu:90-92 */
        free(tr); free(qu); free(sc);
    }
    free(T);
    return 0;
}

int orc_pose_from_cells(const orc_cell *cells, size_t n, const orc_f3 *mp, const orc_f3 *mn,
                        const orc_f3 *sp, const orc_f3 *sn, float d_dist, int cpu_clustering,
                        int use_l1_norm, int use_averaged_clusters, float T_out[16])
{
    return orc_pose_from_cells_w(cells, n, mp, mn, sp, sn, d_dist, cpu_clustering, use_l1_norm,
                                 use_averaged_clusters, 0, T_out);
}
