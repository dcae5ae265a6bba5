/*
 * oslam_host.c -- the C-ABI of include/oslam.h: host orchestration in C over
 * the gfx950 kernels of oslam_kernels.hip.
 *
 * Mirrors the reference's host layer (pcl/alignment/src/cuda/{ppf,model,scene}.cu)
 * with two structural differences: nothing N^2-sized is materialised, and a
 * model's table stays resident in HBM for as many scenes as the caller aligns
 * (the reference rebuilds scene and model per pair, ppf.cu:57-100).
 * There is no CPU fallback: every compute call needs the HIP device.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oslam.h"
#include "oslam_comm.h"
#include "oslam_kernels.h"
#include "oslam_pose.h"
#include "ppf_core.h"

static __thread char g_err[512];
static __thread void *g_stream;

const char *oslam_last_error(void) { return g_err; }

int oslam_set_stream(void *hip_stream)
{
    g_stream = hip_stream;
    return OSLAM_OK;
}

static int fail(int code, const char *what)
{
    if (what != g_err) snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

int oslam_fail(int code, const char *what) { return fail(code, what); }

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                 \
            rc = OSLAM_E_DEVICE;                                                          \
            goto done;                                                                    \
        }                                                                                 \
    } while (0)

#define KCHK(call)                                                                          \
    do {                                                                                    \
        int k_ = (call);                                                                    \
        if (k_ != 0) {                                                                      \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call,                          \
                     hipGetErrorString((hipError_t)k_), __FILE__, __LINE__);                \
            rc = OSLAM_E_DEVICE;                                                            \
            goto done;                                                                      \
        }                                                                                   \
    } while (0)

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------ */
int oslam_params_default(oslam_params *p)
{
    if (!p) return fail(OSLAM_E_INVALID, "params is NULL");
    memset(p, 0, sizeof *p);
    p->ref_point_df = 1;              /* alignment.cpp:134 */
    p->vote_count_threshold = 0.4f;   /* alignment.cpp:136 */
    p->cpu_clustering = 0;
    p->use_l1_norm = 0;
    p->use_averaged_clusters = 0;
    p->dev = 0;
    p->vote_mode = OSLAM_VOTE_EXACT;
    p->shard_rank = 0;
    p->shard_world = 1;
    p->max_cells = 1u << 22;
    p->pose_gpu_min = 0;              /* 0 = default (4096 records) */
    p->no_bucket_spread = 0;
    p->scratch_gib = 0;               /* 0 = default (4 GiB) */
    return OSLAM_OK;
}

int oslam_d_dist_from_cloud(const float *xyz, size_t n, size_t stride_bytes, float tau_d,
                            float *d_dist_out)
{
    float lo[3], hi[3], ext;
    size_t i;
    int a;
    if (!xyz || !d_dist_out || n == 0 || stride_bytes < 12) return fail(OSLAM_E_INVALID, "bad cloud");
    for (a = 0; a < 3; a++) lo[a] = hi[a] = xyz[a];
    for (i = 1; i < n; i++) {
        const float *p = (const float *)((const char *)xyz + i * stride_bytes);
        for (a = 0; a < 3; a++) {
            if (p[a] < lo[a]) lo[a] = p[a];
            if (p[a] > hi[a]) hi[a] = p[a];
        }
    }
    ext = hi[0] - lo[0];
    if (hi[1] - lo[1] > ext) ext = hi[1] - lo[1];
    if (hi[2] - lo[2] > ext) ext = hi[2] - lo[2];
    *d_dist_out = tau_d * ext;        /* alignment.cpp:250-253 */
    return OSLAM_OK;
}

/* ------------------------------------------------------------------------ */
typedef struct cloud_buf {
    int n;
    float *h_xyz, *h_nrm;             /* packed [n][3] host copies (pose stage) */
    float *d_soa;                     /* 6*n floats: px py pz nx ny nz */
    oslamk_cloud k;
} cloud_buf;

struct oslam_model {
    int dev;
    cloud_buf c;
    float d_dist, inv_d_dist;
    oslam_params params;
    oslamk_table table;
    oslamk_entries ent;
    uint32_t n_entries;
    uint64_t num_model_keys;
    float *weights;
    /* align workspace */
    oslamk_counters *d_counters;
    oslamk_cell *d_out;
    uint32_t out_cap;
    oslam_cell *h_out;
    /* multi-GPU: peaks of the last oslam_align_local (in h_out), survivors of this rank (device) */
    size_t n_local;
    uint32_t local_max;
    oslamk_cell *d_union;
    size_t union_cap;
    /* frames T_g of the model points [M][16] and the point weights, for the pose tail on the device */
    float *d_Tm16, *d_weights;
    /* last result: on the host, or still on the device (pose tail ran there) until a tap asks for it */
    oslam_cell *last_cells;
    float *last_poses;
    size_t n_last;
    int last_on_device;
    oslamk_cell *d_pose_cells;
    float *d_pose_T;
    size_t pose_cap;
    /* host copy of the table for the bucket tap */
    oslamk_slot *h_slots;
    /* member of a database group: table.ukeys / reach belong to the group (oslam_db) */
    int shared_union;
    /* its key tables are gone (a database was destroyed without giving them back, or rebuilding them failed):
     * the model can only be destroyed */
    int unusable;
};

struct oslam_scene {
    int dev;
    cloud_buf c;
    float d_dist;
    unsigned df;
    int rank, world;
    int n_ref;
    uint32_t *h_ref_idx, *d_ref_idx;
    float *d_tsg;
    float *d_Ts16;                    /* frames of every reference-point candidate (index % df == 0, all ranks) */
};

static int pick_device(int dev_req, int *dev_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(OSLAM_E_DEVICE, "no HIP device available (the PPF path has no CPU fallback)");
    if (dev_req < 0) dev_req = 0;
    *dev_out = dev_req < n - 1 ? dev_req : n - 1;     /* ppf.cu:45 */
    e = hipSetDevice(*dev_out);
    if (e != hipSuccess) return fail(OSLAM_E_DEVICE, hipGetErrorString(e));
    return OSLAM_OK;
}

/* ---- kept device blocks of the scene path (oslam_kernels.h) ---- */
#define DEVCACHE_SLOTS 64
#define DEVCACHE_MAX_BLOCK ((size_t)1 << 30)        /* larger blocks go straight back to the driver */
typedef struct {
    void *p;
    size_t bytes;
    int dev, used;
} devblock;
static devblock g_devcache[DEVCACHE_SLOTS];
static pthread_mutex_t g_devcache_mu = PTHREAD_MUTEX_INITIALIZER;

int oslam_dev_alloc(void **out, size_t bytes)
{
    int dev = 0, i, best = -1, spare = -1;
    hipError_t e;
    size_t want;
    *out = NULL;
    if (bytes == 0) bytes = 16;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    pthread_mutex_lock(&g_devcache_mu);
    for (i = 0; i < DEVCACHE_SLOTS; i++) {
        devblock *b = &g_devcache[i];
        if (!b->p) { if (spare < 0) spare = i; continue; }
        if (b->used || b->dev != dev || b->bytes < bytes || b->bytes > 2 * bytes + 65536) continue;
        if (best < 0 || b->bytes < g_devcache[best].bytes) best = i;
    }
    if (best >= 0) {
        g_devcache[best].used = 1;
        *out = g_devcache[best].p;
        pthread_mutex_unlock(&g_devcache_mu);
        return 0;
    }
    /* a frame's clouds differ by a few per cent from the last one's: head room lets the next frame reuse the block */
    want = bytes <= DEVCACHE_MAX_BLOCK ? bytes + bytes / 8 + 256 : bytes;
    e = hipMalloc(out, want);
    if (e != hipSuccess) {              /* out of memory with blocks kept: give them back and try once more */
        (void)hipGetLastError();
        for (i = 0; i < DEVCACHE_SLOTS; i++)
            if (g_devcache[i].p && !g_devcache[i].used && g_devcache[i].dev == dev) {
                (void)hipFree(g_devcache[i].p);
                memset(&g_devcache[i], 0, sizeof g_devcache[i]);
                if (spare < 0) spare = i;
            }
        want = bytes;
        e = hipMalloc(out, want);
    }
    if (e == hipSuccess && spare >= 0 && want <= DEVCACHE_MAX_BLOCK) {
        g_devcache[spare].p = *out;
        g_devcache[spare].bytes = want;
        g_devcache[spare].dev = dev;
        g_devcache[spare].used = 1;
    }
    pthread_mutex_unlock(&g_devcache_mu);
    return (int)e;
}

void oslam_dev_free(void *p)
{
    int i;
    if (!p) return;
    pthread_mutex_lock(&g_devcache_mu);
    for (i = 0; i < DEVCACHE_SLOTS; i++)
        if (g_devcache[i].p == p) {
            g_devcache[i].used = 0;
            pthread_mutex_unlock(&g_devcache_mu);
            return;
        }
    pthread_mutex_unlock(&g_devcache_mu);
    (void)hipFree(p);                   /* not one of the kept blocks (table full, or too large) */
}

void oslam_dev_cache_release(int dev)
{
    int i;
    pthread_mutex_lock(&g_devcache_mu);
    for (i = 0; i < DEVCACHE_SLOTS; i++)
        if (g_devcache[i].p && !g_devcache[i].used && g_devcache[i].dev == dev) {
            (void)hipFree(g_devcache[i].p);
            memset(&g_devcache[i], 0, sizeof g_devcache[i]);
        }
    pthread_mutex_unlock(&g_devcache_mu);
}

static void cloud_free(cloud_buf *c)
{
    free(c->h_xyz);
    free(c->h_nrm);
    if (c->d_soa) oslam_dev_free(c->d_soa);
    memset(c, 0, sizeof *c);
}

/* AoS with stride -> packed host copies + SoA in HBM (scene.cu:28-40,68-69) */
static int cloud_upload(cloud_buf *c, const float *xyz, const float *nrm, size_t n, size_t stride)
{
    int rc = OSLAM_OK;
    float *soa = NULL;
    size_t i;
    memset(c, 0, sizeof *c);
    c->n = (int)n;
    c->h_xyz = (float *)malloc(sizeof(float) * 3 * n);
    c->h_nrm = (float *)malloc(sizeof(float) * 3 * n);
    soa = (float *)malloc(sizeof(float) * 6 * n);
    if (!c->h_xyz || !c->h_nrm || !soa) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    for (i = 0; i < n; i++) {
        const float *p = (const float *)((const char *)xyz + i * stride);
        const float *q = (const float *)((const char *)nrm + i * stride);
        int a;
        for (a = 0; a < 3; a++) {
            c->h_xyz[3 * i + a] = p[a];
            c->h_nrm[3 * i + a] = q[a];
            soa[(size_t)a * n + i] = p[a];
            soa[(size_t)(3 + a) * n + i] = q[a];
        }
    }
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&c->d_soa, sizeof(float) * 6 * n));
    HIPCHK(hipMemcpy(c->d_soa, soa, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    c->k.px = c->d_soa;
    c->k.py = c->d_soa + n;
    c->k.pz = c->d_soa + 2 * n;
    c->k.nx = c->d_soa + 3 * n;
    c->k.ny = c->d_soa + 4 * n;
    c->k.nz = c->d_soa + 5 * n;
    c->k.n = (int)n;
done:
    free(soa);
    if (rc != OSLAM_OK) cloud_free(c);
    return rc;
}

/* A cloud that already lies in HBM as [n][6] (x y z nx ny nz per point: what the depth and voxel kernels
 * write): the structure of arrays is made on the device; one copy comes back for the host-side arrays
 * (reference frames, pose stage). */
static int cloud_from_device6(cloud_buf *c, const float *d_aos6, size_t n)
{
    int rc = OSLAM_OK;
    float *h6 = NULL;
    size_t i;
    memset(c, 0, sizeof *c);
    c->n = (int)n;
    c->h_xyz = (float *)malloc(sizeof(float) * 3 * n);
    c->h_nrm = (float *)malloc(sizeof(float) * 3 * n);
    h6 = (float *)malloc(sizeof(float) * 6 * n);
    if (!c->h_xyz || !c->h_nrm || !h6) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&c->d_soa, sizeof(float) * 6 * n));
    KCHK(oslamk_aos6_to_soa(d_aos6, n, c->d_soa, g_stream));
    HIPCHK(hipMemcpyAsync(h6, d_aos6, sizeof(float) * 6 * n, hipMemcpyDeviceToHost, (hipStream_t)g_stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    for (i = 0; i < n; i++) {
        memcpy(c->h_xyz + 3 * i, h6 + 6 * i, 3 * sizeof(float));
        memcpy(c->h_nrm + 3 * i, h6 + 6 * i + 3, 3 * sizeof(float));
    }
    c->k.px = c->d_soa;
    c->k.py = c->d_soa + n;
    c->k.pz = c->d_soa + 2 * n;
    c->k.nx = c->d_soa + 3 * n;
    c->k.ny = c->d_soa + 4 * n;
    c->k.nz = c->d_soa + 5 * n;
    c->k.n = (int)n;
done:
    free(h6);
    if (rc != OSLAM_OK) cloud_free(c);
    return rc;
}

/* ------------------------------------------------------------------------ */
int oslam_voxel_grid(const float *xyz, const float *nrm, size_t n, size_t stride_bytes, float leaf,
                     int dev, float *xyz_out, float *nrm_out, size_t cap, size_t *n_out)
{
    int rc = OSLAM_OK, k, devsel;
    cloud_buf c;
    float *d_out = NULL, *h_out = NULL;
    uint32_t nv = 0;
    size_t i;
    memset(&c, 0, sizeof c);
    if (!xyz || !nrm || !xyz_out || !nrm_out || !n_out || stride_bytes < 12 || !(leaf > 0.0f) || n == 0 ||
        n > 0x7fffffffu)
        return fail(OSLAM_E_INVALID, "bad voxel grid arguments");
    *n_out = 0;
    rc = pick_device(dev, &devsel);
    if (rc != OSLAM_OK) return rc;
    rc = cloud_upload(&c, xyz, nrm, n, stride_bytes);
    if (rc != OSLAM_OK) return rc;
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_out, sizeof(float) * 6 * n));
    k = oslamk_voxel_grid(c.k, leaf, d_out, &nv, g_stream);
    if (k == -1) { rc = fail(OSLAM_E_LIMIT, "leaf size too small for the cloud extent (voxel count overflows int32)"); goto done; }
    if (k != 0) { rc = fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k)); goto done; }
    if (nv > cap) { rc = fail(OSLAM_E_LIMIT, "output capacity too small"); goto done; }
    h_out = (float *)malloc(sizeof(float) * 6 * (nv ? nv : 1));
    if (!h_out) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    if (nv) HIPCHK(hipMemcpy(h_out, d_out, sizeof(float) * 6 * nv, hipMemcpyDeviceToHost));
    for (i = 0; i < nv; i++) {
        memcpy(xyz_out + 3 * i, h_out + 6 * i, 3 * sizeof(float));
        memcpy(nrm_out + 3 * i, h_out + 6 * i + 3, 3 * sizeof(float));
    }
    *n_out = nv;
done:
    free(h_out);
    oslam_dev_free(d_out);
    cloud_free(&c);
    return rc;
}

/* ------------------------------------------------------------------------ */
int oslam_depth_to_cloud(const void *depth, int depth_is_u16, int width, int height, const oslam_camera *cam,
                         int dev, float *xyz_out, float *nrm_out, size_t cap, size_t *n_out)
{
    int rc = OSLAM_OK, k, devsel;
    void *d_img = NULL;
    float *d_out = NULL, *h_out = NULL;
    uint32_t np = 0;
    size_t i, n_pix, px_bytes;
    if (!depth || !cam || !xyz_out || !nrm_out || !n_out || width < 3 || height < 3 || width > 16384 || height > 16384 ||
        !(cam->fx > 0.0f) || !(cam->fy > 0.0f) || !(cam->depth_scale > 0.0f) || !(cam->z_max >= cam->z_min) ||
        !(cam->z_min > 0.0f) || !(cam->max_jump >= 0.0f))
        return fail(OSLAM_E_INVALID, "bad depth image arguments");
    *n_out = 0;
    n_pix = (size_t)width * (size_t)height;
    px_bytes = depth_is_u16 ? 2 : 4;
    rc = pick_device(dev, &devsel);
    if (rc != OSLAM_OK) return rc;
    HIPCHK((hipError_t)oslam_dev_alloc(&d_img, n_pix * px_bytes));
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_out, sizeof(float) * 6 * n_pix));
    HIPCHK(hipMemcpyAsync(d_img, depth, n_pix * px_bytes, hipMemcpyHostToDevice, (hipStream_t)g_stream));
    k = oslamk_depth_to_cloud(d_img, depth_is_u16 != 0, width, height, cam->fx, cam->fy, cam->cx, cam->cy, cam->depth_scale,
                              cam->z_min, cam->z_max, cam->max_jump, d_out, &np, g_stream);
    if (k != 0) { rc = fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k)); goto done; }
    if (np > cap) { rc = fail(OSLAM_E_LIMIT, "output capacity too small"); goto done; }
    h_out = (float *)malloc(sizeof(float) * 6 * (np ? np : 1));
    if (!h_out) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    if (np) HIPCHK(hipMemcpy(h_out, d_out, sizeof(float) * 6 * np, hipMemcpyDeviceToHost));
    for (i = 0; i < np; i++) {
        memcpy(xyz_out + 3 * i, h_out + 6 * i, 3 * sizeof(float));
        memcpy(nrm_out + 3 * i, h_out + 6 * i + 3, 3 * sizeof(float));
    }
    *n_out = np;
done:
    free(h_out);
    oslam_dev_free(d_img);
    oslam_dev_free(d_out);
    return rc;
}

/* ------------------------------------------------------------------------ */
void oslam_model_destroy(oslam_model *m)
{
    if (!m) return;
    (void)hipSetDevice(m->dev);
    cloud_free(&m->c);
    if (m->table.slots) (void)hipFree(m->table.slots);
    if (m->ent.e4) (void)hipFree(m->ent.e4);
    if (m->ent.uv) (void)hipFree(m->ent.uv);
    if (m->ent.pw) (void)hipFree(m->ent.pw);
    if (m->ent.puv) (void)hipFree(m->ent.puv);
    if (m->ent.pdir) (void)hipFree(m->ent.pdir);
    if (m->ent.mi) (void)hipFree(m->ent.mi);
    if (m->table.ukeys && !m->shared_union) (void)hipFree(m->table.ukeys);
    if (m->table.reach && !m->shared_union) (void)hipFree(m->table.reach);
    if (m->table.kmap && !m->shared_union) (void)hipFree(m->table.kmap);
    if (m->table.uids && !m->shared_union) (void)hipFree(m->table.uids);
    if (m->d_counters) (void)hipFree(m->d_counters);
    if (m->d_out) (void)hipFree(m->d_out);
    if (m->d_union) (void)hipFree(m->d_union);
    if (m->table.uinfo) (void)hipFree(m->table.uinfo);
    if (m->d_Tm16) (void)hipFree(m->d_Tm16);
    if (m->d_weights) (void)hipFree(m->d_weights);
    if (m->d_pose_cells) (void)hipFree(m->d_pose_cells);
    if (m->d_pose_T) (void)hipFree(m->d_pose_T);
    free(m->h_out);
    free(m->weights);
    free(m->last_cells);
    free(m->last_poses);
    free(m->h_slots);
    free(m);
}

/* table.uids, table.kmap and table.reach_words from table.ukeys / table.reach (both complete on g_stream): the keys
 * numbered, and the number of every key a reachable distance bin can produce, so that the scene-key kernel looks a
 * pair up with one load.  17^3 words per reachable distance bin (0.8 MB for a model that spans 41 bins). */
static int build_kmap(oslamk_table *t, float d_dist)
{
    int rc = OSLAM_OK;
    uint32_t h_reach[OSLAMK_REACH_BINS / 32], w, top = 0, n_ids = 0, *d_count = NULL;
    t->kmap = NULL;
    t->kmap_bins = 0;
    t->reach_words = 0;
    t->uids = NULL;
    t->n_ids = t->id_bits = t->uinfo_stride = 0;
    /* the keys of the union table numbered 0 .. n_ids-1: what the hit lists carry and sort on, and what the bucket
     * records are indexed by */
    HIPCHK(hipMalloc((void **)&t->uids, sizeof(uint32_t) * (size_t)t->ucap));
    HIPCHK(hipMalloc((void **)&d_count, sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), (hipStream_t)g_stream));
    KCHK(oslamk_union_ids(*t, d_count, g_stream));
    HIPCHK(hipMemcpyAsync(&n_ids, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)g_stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    t->n_ids = n_ids;
    t->id_bits = 1;
    while (t->id_bits < 32u && ((uint64_t)1 << t->id_bits) < (uint64_t)n_ids) t->id_bits++;
    t->uinfo_stride = (n_ids + 63u) & ~63u;
    if (t->uinfo_stride == 0) t->uinfo_stride = 64;
    HIPCHK(hipMemcpy(h_reach, t->reach, sizeof h_reach, hipMemcpyDeviceToHost));
    for (w = 0; w < OSLAMK_REACH_BINS / 32; w++)
        if (h_reach[w]) {
            t->reach_words = w + 1;
            top = 32u * w + (32u - (uint32_t)__builtin_clz(h_reach[w]));    /* highest reachable bin + 1 */
        }
    t->kmap_bins = top < OSLAMK_KMAP_MAX_BINS ? top : OSLAMK_KMAP_MAX_BINS;
    if (t->kmap_bins) {
        HIPCHK(hipMalloc((void **)&t->kmap, sizeof(uint32_t) * (size_t)t->kmap_bins * PC_ANGLE_COMBOS));
        KCHK(oslamk_kmap_build(*t, d_dist, g_stream));
    }
done:
    if (d_count) (void)hipFree(d_count);
    return rc;
}

/* table.ukeys (every distinct key of the model once, at most a quarter full; `distinct` = an upper bound of
 * their number) and table.reach (the distance bins that can produce a key); d_n_keys / d_overflow: device words */
static int build_union(oslam_model *m, uint32_t distinct, uint32_t *d_n_keys, uint32_t *d_overflow)
{
    int rc = OSLAM_OK;
    uint32_t lg = 16;
    while ((1u << lg) < 4u * distinct && lg < OSLAMK_RUN_SHIFT) lg++;
    if ((1u << lg) < 2u * distinct) return fail(OSLAM_E_LIMIT, "more distinct pair keys than the union table can index");
    if (m->table.ukeys && !m->shared_union) (void)hipFree(m->table.ukeys);
    if (m->table.reach && !m->shared_union) (void)hipFree(m->table.reach);
    if (m->table.kmap && !m->shared_union) (void)hipFree(m->table.kmap);
    if (m->table.uids && !m->shared_union) (void)hipFree(m->table.uids);
    m->table.ukeys = NULL;
    m->table.reach = NULL;
    m->table.kmap = NULL;
    m->table.uids = NULL;
    m->shared_union = 0;
    m->table.ucap = 1u << lg;
    m->table.ushift = 32 - lg;
    HIPCHK(hipMalloc((void **)&m->table.ukeys, sizeof(uint32_t) * (size_t)m->table.ucap));
    HIPCHK(hipMemsetAsync(m->table.ukeys, 0, sizeof(uint32_t) * (size_t)m->table.ucap, (hipStream_t)g_stream));
    HIPCHK(hipMemsetAsync(d_overflow, 0, sizeof(uint32_t), (hipStream_t)g_stream));
    HIPCHK(hipMemsetAsync(d_n_keys, 0, sizeof(uint32_t), (hipStream_t)g_stream));
    KCHK(oslamk_union_build(m->table, d_n_keys, d_overflow, g_stream));
    /* which distance bins can produce a model key at all (lets the scene-key kernel drop far pairs) */
    HIPCHK(hipMalloc((void **)&m->table.reach, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32)));
    HIPCHK(hipMemsetAsync(m->table.reach, 0, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32), (hipStream_t)g_stream));
    KCHK(oslamk_reach_build(m->table, m->d_dist, g_stream));
    rc = build_kmap(&m->table, m->d_dist);
done:
    return rc;
}

/* table.uinfo: the bucket of every union-table slot in every slice (what the vote kernel reads) */
static int build_uinfo(oslam_model *m)
{
    int rc = OSLAM_OK;
    const size_t bytes = sizeof(oslamk_uinfo) * (size_t)m->table.n_slices * (size_t)m->table.uinfo_stride;
    if (m->table.uinfo) { (void)hipFree(m->table.uinfo); m->table.uinfo = NULL; }
    HIPCHK(hipMalloc((void **)&m->table.uinfo, bytes));
    HIPCHK(hipMemsetAsync(m->table.uinfo, 0, bytes, (hipStream_t)g_stream));
    KCHK(oslamk_uinfo_build(m->table, g_stream));
done:
    return rc;
}

int oslam_model_create(const float *xyz, const float *nrm, size_t n, size_t stride_bytes,
                       float d_dist, const oslam_params *params, oslam_model **out)
{
    int rc = OSLAM_OK;
    oslam_model *m = NULL;
    uint32_t *d_small = NULL;        /* [0..n_slices) n_unique, then overflow, total, n_first */
    uint32_t h_small[64 + 3];
    float *h_tmg = NULL, *d_tmg = NULL;
    int n_slices, s;
    uint32_t cap;
    size_t n_pairs;

    if (!out) return fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (!xyz || !nrm || stride_bytes < 12 || !(d_dist > 0.0f)) return fail(OSLAM_E_INVALID, "bad model arguments");
    if (n < 2) return fail(OSLAM_E_INVALID, "model needs at least 2 points");
    if (n > 46340) return fail(OSLAM_E_LIMIT, "model larger than 46340 points (32-bit pair index, kernel.cu:433)");
    m = (oslam_model *)calloc(1, sizeof *m);
    if (!m) return fail(OSLAM_E_NOMEM, "host allocation failed");
    if (params) m->params = *params; else oslam_params_default(&m->params);
    if (m->params.max_cells == 0) m->params.max_cells = 1u << 22;
    rc = pick_device(m->params.dev, &m->dev);
    if (rc != OSLAM_OK) goto done;
    rc = cloud_upload(&m->c, xyz, nrm, n, stride_bytes);
    if (rc != OSLAM_OK) goto done;
    m->d_dist = d_dist;
    m->inv_d_dist = 1.0f / d_dist;
    m->weights = (float *)malloc(sizeof(float) * n);
    if (!m->weights) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    for (s = 0; s < (int)n; s++) m->weights[s] = 1.0f;          /* model.cu:67 */

    n_slices = (int)((n + OSLAMK_SLICE - 1) / OSLAMK_SLICE);
    if (n_slices > 64) { rc = fail(OSLAM_E_LIMIT, "too many model slices"); goto done; }
    HIPCHK(hipMalloc((void **)&d_small, sizeof(uint32_t) * (64 + 3)));

    /* pass 1 with table growth: a slice table is kept at most half full */
    for (cap = 1u << 16;; cap <<= 1) {
        int grow = 0;
        uint32_t lg = 0;
        while ((1u << lg) < cap) lg++;
        if (m->table.slots) { (void)hipFree(m->table.slots); m->table.slots = NULL; }
        HIPCHK(hipMalloc((void **)&m->table.slots, sizeof(oslamk_slot) * (size_t)cap * n_slices));
        HIPCHK(hipMemsetAsync(m->table.slots, 0, sizeof(oslamk_slot) * (size_t)cap * n_slices, (hipStream_t)g_stream));
        HIPCHK(hipMemsetAsync(d_small, 0, sizeof(uint32_t) * (64 + 3), (hipStream_t)g_stream));
        m->table.cap = cap;
        m->table.shift = 32 - lg;
        m->table.n_slices = n_slices;
        KCHK(oslamk_model_count(m->c.k, m->d_dist, m->inv_d_dist, m->table, d_small, d_small + 64, g_stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
        HIPCHK(hipMemcpy(h_small, d_small, sizeof h_small, hipMemcpyDeviceToHost));
        if (h_small[64]) grow = 1;
        for (s = 0; s < n_slices; s++) if (h_small[s] > cap / 2) grow = 1;
        if (!grow) break;
        if (cap >= (1u << 26)) { rc = fail(OSLAM_E_LIMIT, "model hash table would exceed 2^26 slots per slice"); goto done; }
    }
    KCHK(oslamk_table_scan(m->table, d_small + 65, g_stream));
    /* union of all slices' keys, kept at most a quarter full */
    {
        uint32_t sum = 0;
        for (s = 0; s < n_slices; s++) sum += h_small[s];
        rc = build_union(m, sum, d_small + 66, d_small + 64);
        if (rc != OSLAM_OK) goto done;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    HIPCHK(hipMemcpy(h_small, d_small, sizeof h_small, hipMemcpyDeviceToHost));
    if (h_small[64]) { rc = fail(OSLAM_E_LIMIT, "union key table overflow"); goto done; }
    m->n_entries = h_small[65];
    m->ent.n_real = m->n_entries;
    m->num_model_keys = (uint64_t)h_small[66] + 1;    /* + the key-0 bucket of the n self pairs */
    n_pairs = m->n_entries ? m->n_entries : 1;

    /* rows y,z of T_m_g per model point, on the host with libm (kernel.cu:310-318) */
    h_tmg = (float *)malloc(sizeof(float) * 8 * n);
    if (!h_tmg) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    oslam_T_g_rows(m->c.h_xyz, m->c.h_nrm, NULL, n, h_tmg);
    HIPCHK(hipMalloc((void **)&d_tmg, sizeof(float) * 8 * n));
    HIPCHK(hipMemcpy(d_tmg, h_tmg, sizeof(float) * 8 * n, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&m->ent.e4, sizeof(uint32_t) * (n_pairs + 256)));   /* + a chunk: the vote kernel loads whole chunks */
    HIPCHK(hipMalloc((void **)&m->ent.mi, sizeof(uint16_t) * n_pairs));
    if (m->params.vote_mode != OSLAM_VOTE_FAST)
        HIPCHK(hipMalloc((void **)&m->ent.uv, sizeof(oslamk_uv) * n_pairs));
    /* every word a padding entry until the fill pass writes it: padding votes into the accumulator's sink row */
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)m->ent.e4, (int)PC_ROW_SINK, n_pairs + 256, (hipStream_t)g_stream));
    KCHK(oslamk_model_fill(m->c.k, m->d_dist, m->inv_d_dist, m->table, d_tmg, m->ent, g_stream));
    if (!m->params.no_bucket_spread) KCHK(oslamk_bucket_spread(m->table, m->ent, g_stream));   /* the switch is for A/B measurements */
    if (m->ent.uv) {
        /* exact mode: every bucket once more in the order of the votes' positions inside their bins, with the uv of
         * its entries (oslamk_entries.pw / .puv); the uv in bucket order are not needed after that */
        HIPCHK(hipMalloc((void **)&m->ent.pw, sizeof(uint32_t) * n_pairs));
        HIPCHK(hipMalloc((void **)&m->ent.puv, sizeof(oslamk_uv) * n_pairs));
        HIPCHK(hipMalloc((void **)&m->ent.pdir, sizeof(uint16_t) * ((n_pairs + 256) << OSLAMK_PDIR_SHIFT)));
        KCHK(oslamk_bucket_psort(m->table, m->ent, g_stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
        (void)hipFree(m->ent.uv);
        m->ent.uv = NULL;
    }
    rc = build_uinfo(m);
    if (rc != OSLAM_OK) goto done;
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));

    m->out_cap = m->params.max_cells;
    HIPCHK(hipMalloc((void **)&m->d_counters, sizeof(oslamk_counters)));
    HIPCHK(hipMalloc((void **)&m->d_out, sizeof(oslamk_cell) * (size_t)m->out_cap));
    m->h_out = (oslam_cell *)malloc(sizeof(oslam_cell) * (size_t)m->out_cap);
    if (!m->h_out) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
done:
    free(h_tmg);
    if (d_tmg) (void)hipFree(d_tmg);
    if (d_small) (void)hipFree(d_small);
    if (rc != OSLAM_OK) { oslam_model_destroy(m); return rc; }
    *out = m;
    return OSLAM_OK;
}

/* ------------------------------------------------------------------------
 * persistent model database: one file per built model
 * ---------------------------------------------------------------------- */
static int build_uinfo(oslam_model *m);

#define OSLAM_DB_MAGIC 0x4c444d4f534c4f00ull     /* "\0OLSOMDL" */
#define OSLAM_DB_VERSION 7u                      /* table layout: 16-B slots, slices of 2046 points, e4 = theta (2^-21 turn) << 11 | half << 10 | row,
                                                  * padding words = row 1023; checksum covers the header;
                                                  * 7: exact mode stores the buckets a second time in vote-position order (pw, puv) instead of uv */
typedef struct db_header {
    uint64_t magic;
    uint32_t version, vote_mode;
    uint32_t n_points, n_slices, cap, shift, ucap, ushift, n_entries, has_uv;
    uint64_t num_model_keys;
    float d_dist, inv_d_dist;
    uint64_t checksum;                           /* FNV-1a 64 over the header (this field zero) and every payload byte, in file order */
} db_header;

static uint64_t fnv64(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    size_t i;
    for (i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

/* device array <-> file through a bounded staging buffer */
static int db_write_dev(FILE *f, const void *dev, size_t bytes, uint64_t *sum)
{
    int rc = OSLAM_OK;
    const size_t chunk = (size_t)64 << 20;
    char *h = (char *)malloc(bytes < chunk ? (bytes ? bytes : 1) : chunk);
    size_t off;
    if (!h) return fail(OSLAM_E_NOMEM, "host allocation failed");
    for (off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        HIPCHK(hipMemcpy(h, (const char *)dev + off, n, hipMemcpyDeviceToHost));
        *sum = fnv64(*sum, h, n);
        if (fwrite(h, 1, n, f) != n) { rc = fail(OSLAM_E_INVALID, "short write"); goto done; }
    }
done:
    free(h);
    return rc;
}

static int db_read_dev(FILE *f, void *dev, size_t bytes, uint64_t *sum)
{
    int rc = OSLAM_OK;
    const size_t chunk = (size_t)64 << 20;
    char *h = (char *)malloc(bytes < chunk ? (bytes ? bytes : 1) : chunk);
    size_t off;
    if (!h) return fail(OSLAM_E_NOMEM, "host allocation failed");
    for (off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        if (fread(h, 1, n, f) != n) { rc = fail(OSLAM_E_INVALID, "model file is truncated"); goto done; }
        *sum = fnv64(*sum, h, n);
        HIPCHK(hipMemcpy((char *)dev + off, h, n, hipMemcpyHostToDevice));
    }
done:
    free(h);
    return rc;
}

static uint32_t log2_exact(uint32_t v)            /* v a power of two */
{
    uint32_t lg = 0;
    while ((1u << lg) < v) lg++;
    return lg;
}

int oslam_model_save(const oslam_model *m, const char *path)
{
    int rc = OSLAM_OK;
    FILE *f = NULL;
    db_header hd;
    uint64_t sum = 0xcbf29ce484222325ull;
    const size_t n = m ? (size_t)m->c.n : 0;
    if (!m || !path) return fail(OSLAM_E_INVALID, "NULL argument");
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    memset(&hd, 0, sizeof hd);
    hd.magic = OSLAM_DB_MAGIC;
    hd.version = OSLAM_DB_VERSION;
    hd.vote_mode = (uint32_t)m->params.vote_mode;
    hd.n_points = (uint32_t)n;
    hd.n_slices = (uint32_t)m->table.n_slices;
    hd.cap = m->table.cap;
    hd.shift = m->table.shift;
    hd.ucap = m->table.ucap;
    hd.ushift = m->table.ushift;
    hd.n_entries = m->n_entries;
    hd.has_uv = m->ent.pw ? 1u : 0u;
    hd.num_model_keys = m->num_model_keys;
    hd.d_dist = m->d_dist;
    hd.inv_d_dist = m->inv_d_dist;
    f = fopen(path, "wb");
    if (!f) return fail(OSLAM_E_INVALID, "cannot open the model file for writing");
    if (fwrite(&hd, sizeof hd, 1, f) != 1) { rc = fail(OSLAM_E_INVALID, "short write"); goto done; }
    /* checksummed: the header (checksum field still zero), host cloud, weights, then the device arrays */
    sum = fnv64(sum, &hd, sizeof hd);
    sum = fnv64(sum, m->c.h_xyz, 12 * n);
    sum = fnv64(sum, m->c.h_nrm, 12 * n);
    sum = fnv64(sum, m->weights, 4 * n);
    if (fwrite(m->c.h_xyz, 12, n, f) != n || fwrite(m->c.h_nrm, 12, n, f) != n || fwrite(m->weights, 4, n, f) != n) {
        rc = fail(OSLAM_E_INVALID, "short write");
        goto done;
    }
    rc = db_write_dev(f, m->table.slots, sizeof(oslamk_slot) * (size_t)hd.cap * hd.n_slices, &sum);
    if (rc == OSLAM_OK) rc = db_write_dev(f, m->table.ukeys, sizeof(uint32_t) * (size_t)hd.ucap, &sum);
    if (rc == OSLAM_OK) rc = db_write_dev(f, m->table.reach, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32), &sum);
    if (rc == OSLAM_OK) rc = db_write_dev(f, m->ent.e4, sizeof(uint32_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK) rc = db_write_dev(f, m->ent.mi, sizeof(uint16_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_write_dev(f, m->ent.pw, sizeof(uint32_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_write_dev(f, m->ent.puv, sizeof(oslamk_uv) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_write_dev(f, m->ent.pdir, sizeof(uint16_t) * ((size_t)hd.n_entries << OSLAMK_PDIR_SHIFT), &sum);
    if (rc != OSLAM_OK) goto done;
    hd.checksum = sum;
    if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&hd, sizeof hd, 1, f) != 1) rc = fail(OSLAM_E_INVALID, "short write");
done:
    if (f && fclose(f) != 0 && rc == OSLAM_OK) rc = fail(OSLAM_E_INVALID, "short write");
    return rc;
}

int oslam_model_load(const char *path, const oslam_params *params, oslam_model **out)
{
    int rc = OSLAM_OK;
    FILE *f = NULL;
    db_header hd, hz;
    oslam_model *m = NULL;
    float *xyz = NULL, *nrm = NULL;
    oslamk_slot *h_slots = NULL;
    uint64_t sum = 0xcbf29ce484222325ull;
    size_t n, n_pairs, n_slots, i;
    if (!out) return fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (!path) return fail(OSLAM_E_INVALID, "path is NULL");
    f = fopen(path, "rb");
    if (!f) return fail(OSLAM_E_INVALID, "cannot open the model file");
    if (fread(&hd, sizeof hd, 1, f) != 1 || hd.magic != OSLAM_DB_MAGIC) { rc = fail(OSLAM_E_INVALID, "not a model file"); goto done; }
    if (hd.version != OSLAM_DB_VERSION) { rc = fail(OSLAM_E_INVALID, "model file has another table layout version"); goto done; }
    /* every field a kernel indexes with is checked against the others: a stale or damaged header must not
     * reach the GPU (slot_of() shifts by `shift`, the vote kernel dereferences uv in exact mode) */
    n = hd.n_points;
    if (n < 2 || n > 46340 || hd.n_slices != (n + OSLAMK_SLICE - 1) / OSLAMK_SLICE || hd.n_slices > 64 ||
        hd.cap < 2 || (hd.cap & (hd.cap - 1)) || hd.cap > (1u << 26) || hd.ucap < 2 || (hd.ucap & (hd.ucap - 1)) ||
        hd.ucap > (1u << OSLAMK_RUN_SHIFT) || hd.shift != 32u - log2_exact(hd.cap) || hd.ushift != 32u - log2_exact(hd.ucap) ||
        !(hd.d_dist > 0.0f) || hd.inv_d_dist != 1.0f / hd.d_dist || hd.has_uv > 1u ||
        hd.vote_mode > (uint32_t)OSLAM_VOTE_FAST || (hd.vote_mode != (uint32_t)OSLAM_VOTE_FAST && !hd.has_uv) ||
        (uint64_t)hd.n_entries > (uint64_t)n * (n - 1) + 3ull * (uint64_t)hd.cap * hd.n_slices ||
        hd.num_model_keys > (uint64_t)hd.ucap + 1) {
        rc = fail(OSLAM_E_INVALID, "model file header is inconsistent");
        goto done;
    }
    hz = hd;
    hz.checksum = 0;
    sum = fnv64(sum, &hz, sizeof hz);
    m = (oslam_model *)calloc(1, sizeof *m);
    if (!m) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    if (params) m->params = *params; else oslam_params_default(&m->params);
    if (m->params.max_cells == 0) m->params.max_cells = 1u << 22;
    if (params && (uint32_t)params->vote_mode != hd.vote_mode && !(hd.has_uv && params->vote_mode == OSLAM_VOTE_FAST)) {
        rc = fail(OSLAM_E_INVALID, "model file was built in fast vote mode: it has no exact entries");
        goto done;
    }
    if (!params) m->params.vote_mode = (int)hd.vote_mode;
    rc = pick_device(m->params.dev, &m->dev);
    if (rc != OSLAM_OK) goto done;
    xyz = (float *)malloc(12 * n);
    nrm = (float *)malloc(12 * n);
    m->weights = (float *)malloc(4 * n);
    n_slots = (size_t)hd.cap * hd.n_slices;
    h_slots = (oslamk_slot *)malloc(sizeof(oslamk_slot) * n_slots);
    if (!xyz || !nrm || !m->weights || !h_slots) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    if (fread(xyz, 12, n, f) != n || fread(nrm, 12, n, f) != n || fread(m->weights, 4, n, f) != n ||
        fread(h_slots, sizeof(oslamk_slot), n_slots, f) != n_slots) {
        rc = fail(OSLAM_E_INVALID, "model file is truncated");
        goto done;
    }
    sum = fnv64(sum, xyz, 12 * n);
    sum = fnv64(sum, nrm, 12 * n);
    sum = fnv64(sum, m->weights, 4 * n);
    sum = fnv64(sum, h_slots, sizeof(oslamk_slot) * n_slots);
    /* the buckets lie one behind the other in slot order, each rounded up to four entries (k_table_scan), and none
     * reaches past the entry arrays: the vote kernel addresses a slice's entries relative to its first slot's start */
    {
        uint64_t run = 0;
        for (i = 0; i < n_slots; i++) {
            if (h_slots[i].start != run || run + h_slots[i].len > hd.n_entries) {
                rc = fail(OSLAM_E_INVALID, "model file: a bucket lies outside the entry arrays or out of order");
                goto done;
            }
            run += ((uint64_t)h_slots[i].len + 3u) & ~(uint64_t)3u;
        }
    }
    rc = cloud_upload(&m->c, xyz, nrm, n, 12);
    if (rc != OSLAM_OK) goto done;
    m->d_dist = hd.d_dist;
    m->inv_d_dist = hd.inv_d_dist;
    m->table.cap = hd.cap;
    m->table.shift = hd.shift;
    m->table.n_slices = (int)hd.n_slices;
    m->table.ucap = hd.ucap;
    m->table.ushift = hd.ushift;
    m->n_entries = hd.n_entries;
    m->ent.n_real = hd.n_entries;
    m->num_model_keys = hd.num_model_keys;
    n_pairs = hd.n_entries ? hd.n_entries : 1;
    HIPCHK(hipMalloc((void **)&m->table.slots, sizeof(oslamk_slot) * n_slots));
    HIPCHK(hipMalloc((void **)&m->table.ukeys, sizeof(uint32_t) * (size_t)hd.ucap));
    HIPCHK(hipMalloc((void **)&m->table.reach, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32)));
    HIPCHK(hipMalloc((void **)&m->ent.e4, sizeof(uint32_t) * (n_pairs + 256)));   /* + a chunk: the vote kernel loads whole chunks */
    HIPCHK(hipMalloc((void **)&m->ent.mi, sizeof(uint16_t) * n_pairs));
    if (hd.has_uv) {
        HIPCHK(hipMalloc((void **)&m->ent.pw, sizeof(uint32_t) * n_pairs));
        HIPCHK(hipMalloc((void **)&m->ent.puv, sizeof(oslamk_uv) * n_pairs));
        HIPCHK(hipMalloc((void **)&m->ent.pdir, sizeof(uint16_t) * ((n_pairs + 256) << OSLAMK_PDIR_SHIFT)));
    }
    HIPCHK(hipMemcpy(m->table.slots, h_slots, sizeof(oslamk_slot) * n_slots, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)m->ent.e4, (int)PC_ROW_SINK, n_pairs + 256, (hipStream_t)g_stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    rc = db_read_dev(f, m->table.ukeys, sizeof(uint32_t) * (size_t)hd.ucap, &sum);
    if (rc == OSLAM_OK) rc = db_read_dev(f, m->table.reach, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32), &sum);
    if (rc == OSLAM_OK) rc = db_read_dev(f, m->ent.e4, sizeof(uint32_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK) rc = db_read_dev(f, m->ent.mi, sizeof(uint16_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_read_dev(f, m->ent.pw, sizeof(uint32_t) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_read_dev(f, m->ent.puv, sizeof(oslamk_uv) * (size_t)hd.n_entries, &sum);
    if (rc == OSLAM_OK && hd.has_uv) rc = db_read_dev(f, m->ent.pdir, sizeof(uint16_t) * ((size_t)hd.n_entries << OSLAMK_PDIR_SHIFT), &sum);
    if (rc != OSLAM_OK) goto done;
    if (sum != hd.checksum) { rc = fail(OSLAM_E_INVALID, "model file checksum mismatch"); goto done; }
    m->h_slots = h_slots;                         /* the bucket tap reads it */
    h_slots = NULL;
    rc = build_kmap(&m->table, m->d_dist);
    if (rc != OSLAM_OK) goto done;
    rc = build_uinfo(m);
    if (rc != OSLAM_OK) goto done;
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    m->out_cap = m->params.max_cells;
    HIPCHK(hipMalloc((void **)&m->d_counters, sizeof(oslamk_counters)));
    HIPCHK(hipMalloc((void **)&m->d_out, sizeof(oslamk_cell) * (size_t)m->out_cap));
    m->h_out = (oslam_cell *)malloc(sizeof(oslam_cell) * (size_t)m->out_cap);
    if (!m->h_out) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
done:
    if (f) fclose(f);
    free(xyz);
    free(nrm);
    free(h_slots);
    if (rc != OSLAM_OK) { oslam_model_destroy(m); return rc; }
    *out = m;
    return OSLAM_OK;
}

int oslam_model_info(const oslam_model *m, size_t *n_points, float *d_dist, uint64_t *table_bytes)
{
    if (!m) return fail(OSLAM_E_INVALID, "NULL handle");
    if (n_points) *n_points = (size_t)m->c.n;
    if (d_dist) *d_dist = m->d_dist;
    if (table_bytes)
        *table_bytes = sizeof(oslamk_slot) * (uint64_t)m->table.cap * (uint64_t)m->table.n_slices +
                       sizeof(uint32_t) * (uint64_t)m->table.ucap + sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32) +
                       sizeof(oslamk_uinfo) * (uint64_t)m->table.uinfo_stride * (uint64_t)m->table.n_slices +
                       (uint64_t)m->n_entries * (4 + 2 + (m->ent.pw ? 12 + (2 << OSLAMK_PDIR_SHIFT) : 0)) + 24ull * (uint64_t)m->c.n;
    return OSLAM_OK;
}

int oslam_model_set_point_weights(oslam_model *m, const float *weights, size_t n)
{
    if (!m || !weights || n != (size_t)m->c.n) return fail(OSLAM_E_INVALID, "bad weights");
    memcpy(m->weights, weights, sizeof(float) * n);
    if (m->d_weights) {
        if (hipSetDevice(m->dev) != hipSuccess ||
            hipMemcpy(m->d_weights, m->weights, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess)
            return fail(OSLAM_E_DEVICE, "cannot update the weights on the device");
    }
    return OSLAM_OK;
}

/* ------------------------------------------------------------------------ */
void oslam_scene_destroy(oslam_scene *s)
{
    if (!s) return;
    (void)hipSetDevice(s->dev);
    cloud_free(&s->c);
    free(s->h_ref_idx);
    oslam_dev_free(s->d_ref_idx);
    oslam_dev_free(s->d_tsg);
    oslam_dev_free(s->d_Ts16);
    free(s);
}

/* Scene::Scene (scene.cu:24-55) from host buffers (xyz != NULL) or from a cloud that lies in HBM as [n][6] */
static int scene_create_any(const float *xyz, const float *nrm, size_t stride_bytes, const float *d_aos6, size_t n,
                            float d_dist, unsigned df, const oslam_params *params, oslam_scene **out)
{
    int rc = OSLAM_OK;
    oslam_scene *s = NULL;
    oslam_params p;
    float *h_tsg = NULL;
    size_t n_all, t;

    *out = NULL;
    if (!(d_dist >= 0.0f) || df == 0) return fail(OSLAM_E_INVALID, "bad scene arguments");
    if (n < 2) return fail(OSLAM_E_INVALID, "scene needs at least 2 points");
    if (n > (1u << 28)) return fail(OSLAM_E_LIMIT, "scene larger than 2^28 points");
    if (params) p = *params; else oslam_params_default(&p);
    if (p.shard_world < 1 || p.shard_rank < 0 || p.shard_rank >= p.shard_world) return fail(OSLAM_E_INVALID, "bad shard");
    s = (oslam_scene *)calloc(1, sizeof *s);
    if (!s) return fail(OSLAM_E_NOMEM, "host allocation failed");
    rc = pick_device(p.dev, &s->dev);
    if (rc != OSLAM_OK) goto done;
    rc = d_aos6 ? cloud_from_device6(&s->c, d_aos6, n) : cloud_upload(&s->c, xyz, nrm, n, stride_bytes);
    if (rc != OSLAM_OK) goto done;
    s->d_dist = d_dist;
    s->df = df;
    s->rank = p.shard_rank;
    s->world = p.shard_world;
    /* reference points: idx % df == 0 (kernel.cu:432), dealt round-robin to ranks */
    n_all = (n + df - 1) / df;
    s->n_ref = 0;
    for (t = (size_t)s->rank; t < n_all; t += (size_t)s->world) s->n_ref++;
    s->h_ref_idx = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_ref ? s->n_ref : 1));
    h_tsg = (float *)malloc(sizeof(float) * 8 * (s->n_ref ? s->n_ref : 1));
    if (!s->h_ref_idx || !h_tsg) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    {
        int k = 0;
        for (t = (size_t)s->rank; t < n_all; t += (size_t)s->world) s->h_ref_idx[k++] = (uint32_t)(t * df);
    }
    oslam_T_g_rows(s->c.h_xyz, s->c.h_nrm, s->h_ref_idx, (size_t)s->n_ref, h_tsg);
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&s->d_ref_idx, sizeof(uint32_t) * (s->n_ref ? s->n_ref : 1)));
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&s->d_tsg, sizeof(float) * 8 * (s->n_ref ? s->n_ref : 1)));
    if (s->n_ref) {
        HIPCHK(hipMemcpy(s->d_ref_idx, s->h_ref_idx, sizeof(uint32_t) * s->n_ref, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->d_tsg, h_tsg, sizeof(float) * 8 * s->n_ref, hipMemcpyHostToDevice));
    }
done:
    free(h_tsg);
    if (rc != OSLAM_OK) { oslam_scene_destroy(s); return rc; }
    *out = s;
    return OSLAM_OK;
}

int oslam_scene_create(const float *xyz, const float *nrm, size_t n, size_t stride_bytes,
                       float d_dist, unsigned df, const oslam_params *params, oslam_scene **out)
{
    if (!out) return fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (!xyz || !nrm || stride_bytes < 12) return fail(OSLAM_E_INVALID, "bad scene arguments");
    return scene_create_any(xyz, nrm, stride_bytes, NULL, n, d_dist, df, params, out);
}

int oslam_scene_from_depth(const void *depth, int depth_is_u16, int width, int height, const oslam_camera *cam,
                           float leaf, float d_dist, unsigned df, const oslam_params *params, oslam_scene **out,
                           size_t *n_points_out)
{
    int rc = OSLAM_OK, k, devsel;
    oslam_params p;
    void *d_img = NULL;
    float *d_pts6 = NULL, *d_soa = NULL, *d_vox6 = NULL;
    const float *d_final;
    uint32_t np = 0, nv = 0;
    size_t n_pix, px_bytes, n_final;
    if (!out) return fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (n_points_out) *n_points_out = 0;
    if (!depth || !cam || width < 3 || height < 3 || width > 16384 || height > 16384 || !(cam->fx > 0.0f) ||
        !(cam->fy > 0.0f) || !(cam->depth_scale > 0.0f) || !(cam->z_max >= cam->z_min) || !(cam->z_min > 0.0f) ||
        !(cam->max_jump >= 0.0f) || !(leaf >= 0.0f))
        return fail(OSLAM_E_INVALID, "bad depth image arguments");
    if (params) p = *params; else oslam_params_default(&p);
    n_pix = (size_t)width * (size_t)height;
    px_bytes = depth_is_u16 ? 2 : 4;
    rc = pick_device(p.dev, &devsel);
    if (rc != OSLAM_OK) return rc;
    HIPCHK((hipError_t)oslam_dev_alloc(&d_img, n_pix * px_bytes));
    HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_pts6, sizeof(float) * 6 * n_pix));
    HIPCHK(hipMemcpyAsync(d_img, depth, n_pix * px_bytes, hipMemcpyHostToDevice, (hipStream_t)g_stream));
    k = oslamk_depth_to_cloud(d_img, depth_is_u16 != 0, width, height, cam->fx, cam->fy, cam->cx, cam->cy, cam->depth_scale,
                              cam->z_min, cam->z_max, cam->max_jump, d_pts6, &np, g_stream);
    if (k != 0) { rc = fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k)); goto done; }
    d_final = d_pts6;
    n_final = np;
    if (leaf > 0.0f && np > 0) {
        oslamk_cloud c;
        HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_soa, sizeof(float) * 6 * (size_t)np));
        HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_vox6, sizeof(float) * 6 * (size_t)np));
        KCHK(oslamk_aos6_to_soa(d_pts6, np, d_soa, g_stream));
        c.px = d_soa; c.py = d_soa + np; c.pz = d_soa + 2 * (size_t)np;
        c.nx = d_soa + 3 * (size_t)np; c.ny = d_soa + 4 * (size_t)np; c.nz = d_soa + 5 * (size_t)np;
        c.n = (int)np;
        k = oslamk_voxel_grid(c, leaf, d_vox6, &nv, g_stream);
        if (k == -1) { rc = fail(OSLAM_E_LIMIT, "leaf size too small for the cloud extent (voxel count overflows int32)"); goto done; }
        if (k != 0) { rc = fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k)); goto done; }
        d_final = d_vox6;
        n_final = nv;
    }
    if (n_final < 2) { rc = fail(OSLAM_E_INVALID, "the depth image leaves fewer than 2 scene points"); goto done; }
    /* the scene's arrays are made from the cloud where it lies; one copy comes back for the host's reference frames */
    rc = scene_create_any(NULL, NULL, 0, d_final, n_final, d_dist, df, &p, out);
    if (rc == OSLAM_OK && n_points_out) *n_points_out = n_final;
done:
    oslam_dev_free(d_img);
    oslam_dev_free(d_pts6);
    oslam_dev_free(d_soa);
    oslam_dev_free(d_vox6);
    return rc;
}

/* ------------------------------------------------------------------------ */
static int check_pair(const oslam_model *m, const oslam_scene *s)
{
    if (!m || !s) return fail(OSLAM_E_INVALID, "NULL handle");
    if (m->unusable) return fail(OSLAM_E_INVALID, "this model lost its key tables with its database: it can only be destroyed");
    if (m->dev != s->dev) return fail(OSLAM_E_INVALID, "model and scene live on different devices");
    /* d_dist 0 = a scene for models of any d_dist: nothing a scene holds here depends on it */
    if (s->d_dist != 0.0f && m->d_dist != s->d_dist) return fail(OSLAM_E_INVALID, "scene d_dist differs from the model's (ppf.cu:64-67)");
    return OSLAM_OK;
}

/* Scratch for the hit lists of one batch of reference points: one pool per device, shared by all
 * models and only live inside a call (calls on one device are serialised by the pool's lock).  The
 * lists are sized by demand: a counting kernel gives, per reference point, the number of scene pairs
 * that can reach a model key at all (an upper bound of its hits, 16 % above them on the bench scene);
 * the host turns the counts into offsets and cuts the reference points into batches that fit the
 * pool.  The pool grows to what a call needs, up to OSLAM_SCRATCH_GIB GiB (default 4; a single
 * reference point that needs more still gets it); oslam_release_scratch frees it. */
#define MAX_DEVICES 64
#define MAX_BATCH_EVENTS 64
#define SLOT_BYTES (sizeof(oslamk_pay) * 2 + sizeof(oslamk_run) + sizeof(uint32_t))
typedef struct {
    pthread_mutex_t lock;
    char *buf;                         /* hit arrays of one batch */
    size_t bytes;
    uint32_t *d_counts;                /* keep_count[cap], hit_count[cap], run_count[cap], hit_off[cap + 1 + batches] */
    uint32_t *h_counts;                /* host staging: keep counts, then offsets */
    size_t counts_cap;
    hipEvent_t ev[4 + 3 * MAX_BATCH_EVENTS];
    int have_events;
    char *d_cluster;                   /* workspace of cluster_scores_on_device */
    size_t cluster_bytes;
    uint32_t *d_redo;                  /* vote workgroups of a batch whose 16-bit counters overflowed */
    oslamk_vote_args *d_vargs, *h_vargs;   /* a group's vote arguments, one per member (h: pinned), for the one-grid launch */
    size_t vargs_cap;
    size_t redo_cap;
} scratch_pool;
static scratch_pool g_pool[MAX_DEVICES];
static pthread_once_t g_pool_once = PTHREAD_ONCE_INIT;

static void pool_init_all(void)
{
    int i;
    for (i = 0; i < MAX_DEVICES; i++) pthread_mutex_init(&g_pool[i].lock, NULL);
}

/* the pool of a device, locked: every entry point that launches on the device holds it for the call */
static scratch_pool *pool_lock(int dev)
{
    pthread_once(&g_pool_once, pool_init_all);
    if (dev < 0 || dev >= MAX_DEVICES) return NULL;
    pthread_mutex_lock(&g_pool[dev].lock);
    return &g_pool[dev];
}

static void pool_unlock(scratch_pool *p)
{
    if (p) pthread_mutex_unlock(&p->lock);
}

int oslam_release_scratch(int dev)
{
    scratch_pool *p = pool_lock(dev);
    int i;
    if (!p) return fail(OSLAM_E_INVALID, "device ordinal out of range");
    if (p->buf || p->d_counts || p->have_events || p->d_cluster || p->d_redo || p->d_vargs || p->h_vargs) {
        if (hipSetDevice(dev) != hipSuccess) { pool_unlock(p); return fail(OSLAM_E_DEVICE, "hipSetDevice failed"); }
        if (p->buf) (void)hipFree(p->buf);
        if (p->d_counts) (void)hipFree(p->d_counts);
        if (p->d_cluster) (void)hipFree(p->d_cluster);
        if (p->d_redo) (void)hipFree(p->d_redo);
        if (p->d_vargs) (void)hipFree(p->d_vargs);
        if (p->h_vargs) (void)hipHostFree(p->h_vargs);
        oslamk_pose_release();
        if (p->have_events)
            for (i = 0; i < 4 + 3 * MAX_BATCH_EVENTS; i++) (void)hipEventDestroy(p->ev[i]);
    }
    if (hipSetDevice(dev) == hipSuccess) oslam_dev_cache_release(dev);   /* the kept blocks of the scene path */
    free(p->h_counts);
    p->buf = NULL;
    p->bytes = 0;
    p->d_counts = NULL;
    p->h_counts = NULL;
    p->counts_cap = 0;
    p->have_events = 0;
    p->d_cluster = NULL;
    p->cluster_bytes = 0;
    p->d_redo = NULL;
    p->redo_cap = 0;
    p->d_vargs = NULL;
    p->h_vargs = NULL;
    p->vargs_cap = 0;
    pool_unlock(p);
    return OSLAM_OK;
}

static size_t scratch_limit(const oslam_model *m)
{
    return (size_t)(m && m->params.scratch_gib > 0 ? m->params.scratch_gib : 4) << 30;
}

/* per-reference counters for n_ref reference points, events */
static int pool_reserve_counts(scratch_pool *p, size_t n_ref)
{
    int rc = OSLAM_OK, i;
    if (!p->have_events) {
        for (i = 0; i < 4 + 3 * MAX_BATCH_EVENTS; i++) HIPCHK(hipEventCreate(&p->ev[i]));
        p->have_events = 1;
    }
    if (p->counts_cap < n_ref) {
        const size_t cap = n_ref + n_ref / 4 + 64;
        if (p->d_counts) { (void)hipFree(p->d_counts); p->d_counts = NULL; }
        free(p->h_counts);
        p->h_counts = NULL;
        p->counts_cap = 0;
        /* offsets: one more than reference points per batch; a batch holds at least one reference point */
        HIPCHK(hipMalloc((void **)&p->d_counts, sizeof(uint32_t) * (5 * cap + 2)));
        p->h_counts = (uint32_t *)malloc(sizeof(uint32_t) * (3 * cap + 2));
        if (!p->h_counts) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
        p->counts_cap = cap;
    }
done:
    return rc;
}

static int pool_reserve_slots(scratch_pool *p, size_t slots, size_t limit)
{
    const size_t base = (slots ? slots : 1) * SLOT_BYTES + 1024;      /* + a wave of hits: the vote kernel loads 64 at a time */
    size_t want = base;
    if (p->bytes >= want) return OSLAM_OK;
    if (p->buf) { (void)hipFree(p->buf); p->buf = NULL; p->bytes = 0; }
    want += want / 8;                 /* head room: the next scene is rarely the same size */
    /* batches are cut to the limit: no head room beyond it, but never less than the batch itself needs (a pool
     * below `base` would be freed and mapped again by every registration) */
    if (want > limit + 1024) want = base > limit + 1024 ? base : limit + 1024;
    if (hipMalloc((void **)&p->buf, want) != hipSuccess) {
        (void)hipGetLastError();
        p->buf = NULL;
        want = (slots ? slots : 1) * SLOT_BYTES + 1024;
        if (hipMalloc((void **)&p->buf, want) != hipSuccess) {
            (void)hipGetLastError();
            p->buf = NULL;
            return fail(OSLAM_E_NOMEM, "no device memory for the hit lists");
        }
    }
    p->bytes = want;
    return OSLAM_OK;
}

/* the arrays of a batch with `slots` places inside the pool */
static void carve_scratch(oslamk_vote_args *a, const scratch_pool *p, size_t slots)
{
    char *b = p->buf;
    a->redo = p->d_redo;
    a->hit_pay = (oslamk_pay *)b;
    b += slots * sizeof(oslamk_pay);
    a->hit_sorted = (oslamk_pay *)b;
    b += slots * sizeof(oslamk_pay);
    a->runs = (oslamk_run *)b;
    b += slots * sizeof(oslamk_run);
    a->hit_key = (uint32_t *)b;
}

/* The batch that starts at reference point `first`: as many reference points as fit `limit_slots` places
 * (at least one).  A list gets its count rounded up to even, so that every list starts 8-byte aligned in
 * the 4-byte key array too.  off (may be NULL) receives the n + 1 offsets; *slots the batch total. */
static int batch_extent(const uint32_t *keep, int first, int n_ref, size_t limit_slots, uint32_t *off, size_t *slots)
{
    size_t t = 0;
    int n = 0;
    while (first + n < n_ref) {
        const size_t need = ((size_t)keep[first + n] + 1u) & ~(size_t)1u;
        if (n > 0 && (t + need > limit_slots || t + need > 0xfffffff0u)) break;
        if (off) off[n] = (uint32_t)t;
        t += need;
        n++;
    }
    if (off) off[n] = (uint32_t)t;
    *slots = t;
    return n;
}

/* The kernels of one registration (or of one reference point for the accumulator tap): count, then per
 * batch scene keys -> hit sort -> votes.  d_ref_idx / d_tsg: the reference points and their frame rows.
 * The caller holds the pool of the device. */
static int run_votes_group(scratch_pool *pool, oslam_model *const *ms, int nm, oslam_scene *s, const uint32_t *d_ref_idx,
                           const float *d_tsg, int n_ref, uint32_t fixed_gmax, uint32_t *acc_dump,
                           oslamk_counters *cnt, float *ms_out, float *ms_vote_kernel, float *ms_key_kernel,
                           uint32_t *launches, uint64_t *probed)
{
    /* ms[0..nm): models that share one union table and d_dist (a database group, or one model): the scene
     * pass -- count, keys, hit sort -- runs once for all of them, then each model votes with its own buckets.
     * cnt[nm]; the vote-kernel time is the sum over the models. */
    oslam_model *m = ms[0];
    int rc = OSLAM_OK, first, nb = 0, i, j;
    oslamk_vote_args a;
    hipStream_t st = (hipStream_t)g_stream;
    hipEvent_t *ev;
    const size_t limit_slots = scratch_limit(ms[0]) / SLOT_BYTES;
    size_t cap, max_batch_slots = 0, redo_stride = 0;
    int one_grid = 0;
    uint32_t *h_keep, *h_off, *d_keep, *d_hitc, *d_runc, *d_off;
    float k0 = 0.0f;
    rc = pool_reserve_counts(pool, (size_t)(n_ref > 0 ? n_ref : 1));
    if (rc != OSLAM_OK) return rc;
    {
        /* one place per vote workgroup of the largest launch: (reference points padded to 8) x slices */
        size_t nsl = 1, need;
        for (j = 0; j < nm; j++) if ((size_t)ms[j]->table.n_slices > nsl) nsl = (size_t)ms[j]->table.n_slices;
        need = (((size_t)(n_ref > 0 ? n_ref : 1) + 7) / 8 * 8) * nsl;
        redo_stride = need;
        if (nm > 1) need *= (size_t)nm;           /* a group voted in one grid: every member its own list */
        if (pool->redo_cap < need) {
            if (pool->d_redo) { (void)hipFree(pool->d_redo); pool->d_redo = NULL; pool->redo_cap = 0; }
            HIPCHK(hipMalloc((void **)&pool->d_redo, sizeof(uint32_t) * (need + need / 4)));
            pool->redo_cap = need + need / 4;
        }
    }
    ev = pool->ev;
    cap = pool->counts_cap;
    d_keep = pool->d_counts;
    d_hitc = d_keep + cap;
    d_runc = d_hitc + cap;
    d_off = d_runc + cap;                       /* [2 * cap + 2] */
    h_keep = pool->h_counts;
    h_off = h_keep + cap;
    memset(&a, 0, sizeof a);
    a.scene = s->c.k;
    a.ref_idx = d_ref_idx;
    a.tsg = d_tsg;
    a.n_ref = n_ref;
    a.d_dist = m->d_dist;
    a.inv_d_dist = m->inv_d_dist;
    a.table = m->table;
    a.ent = m->ent;
    a.thresh = m->params.vote_count_threshold;
    a.fixed_gmax = fixed_gmax;
    a.counters = m->d_counters;
    a.out = m->d_out;
    a.out_cap = m->out_cap;
    a.acc_dump = acc_dump;
    a.dump_ref = acc_dump ? 0 : -1;
    a.mode = (m->params.vote_mode == OSLAM_VOTE_FAST) ? 1 : 0;
    for (j = 0; j < nm; j++) HIPCHK(hipMemsetAsync(ms[j]->d_counters, 0, sizeof(oslamk_counters), st));
    HIPCHK(hipEventRecord(ev[0], st));
    /* 1. demand: pairs within reach, per reference point */
    if (n_ref > 0) {
        HIPCHK(hipMemsetAsync(d_keep, 0, sizeof(uint32_t) * (size_t)n_ref, st));
        a.first_ref = 0;
        a.n_launch = n_ref;
        a.keep_count = d_keep;
        KCHK(oslamk_scene_count(&a, g_stream));
        HIPCHK(hipEventRecord(ev[2], st));
        HIPCHK(hipMemcpyAsync(h_keep, d_keep, sizeof(uint32_t) * (size_t)n_ref, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipEventElapsedTime(&k0, ev[0], ev[2]));
        if (ms_key_kernel) *ms_key_kernel += k0;
        if (probed) {
            uint64_t t = 0;
            for (i = 0; i < n_ref; i++) t += h_keep[i];
            *probed = t;
        }
    }
    /* 2. batches that fit the pool; the offsets of every batch start at 0.  h_off holds, batch after
     * batch, the n + 1 offsets of its n reference points */
    {
        size_t pos = 0;
        for (first = 0; first < n_ref;) {
            size_t slots;
            const int n = batch_extent(h_keep, first, n_ref, limit_slots, h_off + pos, &slots);
            if (slots > max_batch_slots) max_batch_slots = slots;
            pos += (size_t)n + 1;
            first += n;
        }
        rc = pool_reserve_slots(pool, max_batch_slots, scratch_limit(ms[0]));
        if (rc != OSLAM_OK) goto done;
        if (pos) HIPCHK(hipMemcpyAsync(d_off, h_off, sizeof(uint32_t) * pos, hipMemcpyHostToDevice, st));
    }
    /* A group whose frame is one batch votes in ONE grid (k_vote_group): fifty small models are fifty grids of little
     * more than one round of workgroups otherwise, each with its own tail and its own three launches. */
    if (nm > 1 && n_ref > 0 && !acc_dump) {
        size_t slots;
        size_t nsl = 1;
        for (j = 0; j < nm; j++) if ((size_t)ms[j]->table.n_slices > nsl) nsl = (size_t)ms[j]->table.n_slices;
        /* ... where it pays: members whose own grid is a few rounds of workgroups at most.  A member with tens of
         * thousands of workgroups fills the chip by itself, and the kernel that takes its arguments from memory keeps
         * more of them in registers than the one that gets them as kernel arguments (10 x 5000 points against 100k:
         * 342 ms in one grid, 313 ms in ten). */
        one_grid = batch_extent(h_keep, 0, n_ref, limit_slots, NULL, &slots) == n_ref &&
                   ((size_t)n_ref + 7) / 8 * 8 * nsl <= 2048;
        for (j = 1; j < nm && one_grid; j++)
            if ((ms[j]->params.vote_mode == OSLAM_VOTE_FAST) != (ms[0]->params.vote_mode == OSLAM_VOTE_FAST)) one_grid = 0;
        if (one_grid && pool->vargs_cap < (size_t)nm) {
            const size_t want = (size_t)nm + (size_t)nm / 2 + 8;
            if (pool->d_vargs) { (void)hipFree(pool->d_vargs); pool->d_vargs = NULL; }
            if (pool->h_vargs) { (void)hipHostFree(pool->h_vargs); pool->h_vargs = NULL; }
            pool->vargs_cap = 0;
            HIPCHK(hipMalloc((void **)&pool->d_vargs, sizeof(oslamk_vote_args) * want));
            HIPCHK(hipHostMalloc((void **)&pool->h_vargs, sizeof(oslamk_vote_args) * want, hipHostMallocDefault));
            pool->vargs_cap = want;
        }
    }
    /* 3. the batches */
    {
        size_t pos = 0;
        for (first = 0; first < n_ref; nb++) {
            const int timed = nb < MAX_BATCH_EVENTS;
            size_t slots;
            const int n = batch_extent(h_keep, first, n_ref, limit_slots, NULL, &slots);
            a.first_ref = first;
            a.n_launch = n;
            a.keep_count = NULL;
            a.hit_off = d_off + pos;
            a.hit_count = d_hitc;
            a.run_count = d_runc;
            carve_scratch(&a, pool, slots);
            HIPCHK(hipMemsetAsync(d_hitc, 0, sizeof(uint32_t) * (size_t)n, st));
            if (timed) HIPCHK(hipEventRecord(ev[4 + 3 * nb], st));
            KCHK(oslamk_scene_hits(&a, g_stream));
            KCHK(oslamk_sort_hits(&a, g_stream));
            if (timed) HIPCHK(hipEventRecord(ev[4 + 3 * nb + 1], st));
            for (j = 0; j < nm; j++) {
                const oslam_model *mj = ms[j];
                a.table.uinfo = mj->table.uinfo;          /* its buckets, under the shared union slots */
                a.table.n_slices = mj->table.n_slices;
                a.table.slots = mj->table.slots;
                a.table.cap = mj->table.cap;
                a.ent = mj->ent;
                a.thresh = mj->params.vote_count_threshold;
                a.counters = mj->d_counters;
                a.out = mj->d_out;
                a.out_cap = mj->out_cap;
                a.mode = (mj->params.vote_mode == OSLAM_VOTE_FAST) ? 1 : 0;
                if (one_grid) {
                    a.redo = pool->d_redo + (size_t)j * redo_stride;
                    pool->h_vargs[j] = a;
                    continue;
                }
                KCHK(oslamk_vote(&a, g_stream));
                /* the redo list belongs to this launch */
                HIPCHK(hipMemsetAsync(&mj->d_counters->redo_count, 0, sizeof(uint32_t), st));
            }
            if (one_grid) {
                HIPCHK(hipMemcpyAsync(pool->d_vargs, pool->h_vargs, sizeof(oslamk_vote_args) * (size_t)nm, hipMemcpyHostToDevice, st));
                KCHK(oslamk_vote_group(pool->d_vargs, pool->h_vargs, nm, g_stream));
            }
            if (timed) HIPCHK(hipEventRecord(ev[4 + 3 * nb + 2], st));
            if (launches) *launches += 1;
            pos += (size_t)n + 1;
            first += n;
        }
    }
    HIPCHK(hipEventRecord(ev[1], st));
    for (j = 0; j < nm; j++) HIPCHK(hipMemcpyAsync(&cnt[j], ms[j]->d_counters, sizeof *cnt, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (one_grid) {
        /* the one-grid launch leaves the re-vote of workgroups whose 16-bit counters overflowed to here: a member that
         * has any (large planes in a small model: next to never) gets its two passes now and its counters again */
        int again = 0;
        for (j = 0; j < nm; j++)
            if (cnt[j].redo_count) {
                KCHK(oslamk_vote_wide(&pool->h_vargs[j], g_stream));
                HIPCHK(hipMemcpyAsync(&cnt[j], ms[j]->d_counters, sizeof *cnt, hipMemcpyDeviceToHost, st));
                again = 1;
            }
        if (again) HIPCHK(hipStreamSynchronize(st));
    }
    for (j = 0; j < nm; j++)
        if (cnt[j].list_overflow) {
            rc = fail(OSLAM_E_DEVICE, cnt[j].list_overflow & 1u
                          ? "a hit list overflowed: the counting pass and the hit pass disagreed on the pairs within reach"
                          : cnt[j].list_overflow & 2u ? "near-edge search: a hit's key number lies outside the model's bucket records"
                          : cnt[j].list_overflow & 4u ? "near-edge search: a bucket lies outside the model's entry arrays"
                                                      : "near-edge search: a directory place lies outside its bucket segment");
            goto done;
        }
    if (ms_out) HIPCHK(hipEventElapsedTime(ms_out, ev[0], ev[1]));
    for (i = 0; i < nb && i < MAX_BATCH_EVENTS; i++) {
        float k = 0.0f, v = 0.0f;
        HIPCHK(hipEventElapsedTime(&k, ev[4 + 3 * i], ev[4 + 3 * i + 1]));
        HIPCHK(hipEventElapsedTime(&v, ev[4 + 3 * i + 1], ev[4 + 3 * i + 2]));
        if (ms_key_kernel) *ms_key_kernel += k;
        if (ms_vote_kernel) *ms_vote_kernel += v;
    }
done:
    return rc;
}

static int run_votes_refs(scratch_pool *pool, oslam_model *m, oslam_scene *s, const uint32_t *d_ref_idx,
                          const float *d_tsg, int n_ref, uint32_t fixed_gmax, uint32_t *acc_dump,
                          oslamk_counters *cnt, float *ms_out, float *ms_vote_kernel, float *ms_key_kernel,
                          uint32_t *launches, uint64_t *probed)
{
    oslam_model *one[1];
    one[0] = m;
    return run_votes_group(pool, one, 1, s, d_ref_idx, d_tsg, n_ref, fixed_gmax, acc_dump, cnt, ms_out, ms_vote_kernel,
                           ms_key_kernel, launches, probed);
}

static int run_votes(scratch_pool *pool, oslam_model *m, oslam_scene *s, uint32_t fixed_gmax, oslamk_counters *cnt,
                     float *ms_out, float *ms_vote_kernel, float *ms_key_kernel, uint32_t *launches, uint64_t *probed)
{
    return run_votes_refs(pool, m, s, s->d_ref_idx, s->d_tsg, s->n_ref, fixed_gmax, NULL, cnt, ms_out, ms_vote_kernel,
                          ms_key_kernel, launches, probed);
}

/* record buffers (device and host) for at least `need` records; the contents are dropped */
static int grow_records(oslam_model *m, uint64_t need)
{
    oslamk_cell *d_new = NULL;
    oslam_cell *h_new;
    if (need > ((uint64_t)1 << 28)) return fail(OSLAM_E_LIMIT, "more than 2^28 accumulator peaks above the threshold");
    if (hipMalloc((void **)&d_new, sizeof(oslamk_cell) * need) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OSLAM_E_NOMEM, "no device memory for the accumulator peaks");
    }
    h_new = (oslam_cell *)malloc(sizeof(oslam_cell) * need);
    if (!h_new) { (void)hipFree(d_new); return fail(OSLAM_E_NOMEM, "host allocation failed"); }
    (void)hipFree(m->d_out);
    free(m->h_out);
    m->d_out = d_new;
    m->h_out = h_new;
    m->out_cap = (uint32_t)need;
    m->n_local = 0;
    return OSLAM_OK;
}

static int ensure_union(oslam_model *m, size_t n)
{
    if (m->union_cap >= n) return OSLAM_OK;
    if (m->d_union) { (void)hipFree(m->d_union); m->d_union = NULL; m->union_cap = 0; }
    if (hipMalloc((void **)&m->d_union, sizeof(oslamk_cell) * (n + n / 4)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OSLAM_E_NOMEM, "no device memory for the accumulator peaks");
    }
    m->union_cap = n + n / 4;
    return OSLAM_OK;
}

/* vote + D2H of emitted cells; handles an overflowing record buffer by a second,
 * exactly thresholded launch */
static int vote_and_fetch(scratch_pool *pool, oslam_model *m, oslam_scene *s, oslamk_counters *cnt, size_t *n_cells,
                          oslam_stats *st, size_t leave_on_device_from)
{
    int rc = OSLAM_OK;
    float ms = 0.0f, ms2 = 0.0f, msv = 0.0f, msk = 0.0f;
    uint32_t launches = 0;
    uint64_t probed = 0;
    rc = run_votes(pool, m, s, 0, cnt, &ms, &msv, &msk, &launches, &probed);
    if (rc != OSLAM_OK) return rc;
    if (cnt->out_count > m->out_cap) {
        uint32_t g = cnt->gmax;
        rc = run_votes(pool, m, s, g, cnt, &ms2, &msv, &msk, &launches, &probed);
        if (rc != OSLAM_OK) return rc;
        cnt->gmax = g;
        if (cnt->out_count > m->out_cap) {
            /* even the exactly thresholded set is larger than the record buffer: the count is known now, so
             * the buffers grow to it (up to 2^28 records = 4 GiB) and the launch is repeated */
            rc = grow_records(m, (uint64_t)cnt->out_count + cnt->out_count / 8 + 1024);
            if (rc != OSLAM_OK) return rc;
            rc = run_votes(pool, m, s, g, cnt, &ms2, &msv, &msk, &launches, &probed);
            if (rc != OSLAM_OK) return rc;
            cnt->gmax = g;
            if (cnt->out_count > m->out_cap)
                return fail(OSLAM_E_LIMIT, "more accumulator peaks than the record buffer after growing it");
        }
    }
    if (getenv("OSLAM_PROF"))      /* only a -DVOTE_PROF build fills these */
        fprintf(stderr, "[oslam prof] k_vote wave cycles: pre-scan %llu, voting %llu, wait at the barrier behind it %llu, "
                        "of the voting: near-edge search %llu\n",
                cnt->prof[0], cnt->prof[1], cnt->prof[2], cnt->prof[3]);
    *n_cells = cnt->out_count;
    /* the records stay in HBM when the pose tail runs there (leave_on_device_from = its lower bound, 0 = never) */
    if (*n_cells && !(leave_on_device_from && *n_cells >= leave_on_device_from)) {
        HIPCHK(hipMemcpy(m->h_out, m->d_out, sizeof(oslam_cell) * *n_cells, hipMemcpyDeviceToHost));
    }
    if (st) {
        st->num_scene_ppfs = (uint64_t)s->n_ref * (uint64_t)(s->c.n - 1);
        st->num_hits = cnt->hits;
        st->num_votes = cnt->votes;
        st->num_unique_votes = cnt->nonzero_cells;
        st->num_model_keys = m->num_model_keys;
        st->max_count = cnt->gmax;
        st->num_emitted = cnt->out_count;
        st->ms_vote = ms + ms2;
        st->ms_vote_kernel = msv;
        st->ms_key_kernel = msk;
        st->vote_launches = launches;
        st->num_pairs_probed = probed;
        st->num_entries_streamed = cnt->entries;
        st->num_items = cnt->items;
        st->wide_workgroups = cnt->redo_total;
        st->scratch_bytes = pool->bytes;
    }
done:
    return rc;
}

/* the pool the running call holds (the clustering hook below has no other way to reach it) */
static __thread scratch_pool *g_cur_pool;

/* clustering scores on the bound device (see oslam_pose.h); any failure makes the host loop run */
static int cluster_scores_on_device(size_t n, const float *trans, const float *quat, const float *wv,
                                    const int32_t *cell, const uint32_t *hash_idx, float d_dist, int use_l1,
                                    float *score)
{
    int rc = OSLAM_OK;
    char *d = NULL, *h = NULL;
    /* pose order: cell [n][3]; sorted order: hash [n], pose index [n], quat, trans, votes; out: score (pose order) */
    const size_t o_c = 0, o_sh = o_c + 12 * n, o_si = o_sh + 4 * n, o_sq = o_si + 4 * n,
                 o_st = o_sq + 16 * n, o_sw = o_st + 12 * n, o_sc = o_sw + 4 * n, o_tab = o_sc + 4 * n,
                 total = o_tab + 4 * oslamk_cluster_table_words((int)n);
    size_t j;
    int whole = 1;
    uint64_t whole_sum = 0;
    hipStream_t st = (hipStream_t)g_stream;
    h = (char *)malloc(o_sc);
    if (!h) return OSLAM_E_NOMEM;
    memcpy(h + o_c, cell, 12 * n);
    for (j = 0; j < n; j++) {
        const uint32_t o = hash_idx[2 * j + 1];
        ((uint32_t *)(h + o_sh))[j] = hash_idx[2 * j];
        ((uint32_t *)(h + o_si))[j] = o;
        memcpy(h + o_sq + 16 * j, quat + 4 * o, 16);
        memcpy(h + o_st + 12 * j, trans + 3 * o, 12);
        ((float *)(h + o_sw))[j] = wv[o];
        /* whole numbers with a sum below 2^24: any order of adding them gives the same float (oslamk_cluster_scores) */
        if (whole && wv[o] >= 0.0f && wv[o] < 16777216.0f && wv[o] == (float)(uint32_t)wv[o]) whole_sum += (uint32_t)wv[o];
        else whole = 0;
    }
    if (whole_sum >= (1u << 24) - 1u) whole = 0;
    /* persistent workspace in the device's pool (the caller holds its lock) */
    if (!g_cur_pool) { rc = OSLAM_E_DEVICE; goto done; }
    if (g_cur_pool->cluster_bytes < total) {
        if (g_cur_pool->d_cluster) (void)hipFree(g_cur_pool->d_cluster);
        g_cur_pool->d_cluster = NULL;
        g_cur_pool->cluster_bytes = 0;
        HIPCHK(hipMalloc((void **)&g_cur_pool->d_cluster, total + total / 4));
        g_cur_pool->cluster_bytes = total + total / 4;
    }
    d = g_cur_pool->d_cluster;
    HIPCHK(hipMemcpyAsync(d, h, o_sc, hipMemcpyHostToDevice, st));
    KCHK(oslamk_cluster_scores((int)n, (const int *)(d + o_c), (const uint32_t *)(d + o_sh), (const uint32_t *)(d + o_si),
                               (const float *)(d + o_sq), (const float *)(d + o_st),
                               (const float *)(d + o_sw), d_dist, use_l1, (float *)(d + o_sc), whole, NULL, (uint32_t *)(d + o_tab), g_stream));
    HIPCHK(hipMemcpyAsync(score, d + o_sc, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
done:
    free(h);
    return rc;
}

/* ---- pose tail on the device (oslam_posegpu.hip) for large peak sets ---- */

/* 0: the tail may run on the device; the host-only variants keep the host path */
static size_t pose_gpu_from(const oslam_model *m)
{
    if (m->params.cpu_clustering || m->params.use_averaged_clusters) return 0;
    return m->params.pose_gpu_min ? (size_t)m->params.pose_gpu_min : 4096;      /* tests force either tail through the parameter */
}

static void drop_last(oslam_model *m)
{
    free(m->last_cells);
    free(m->last_poses);
    m->last_cells = NULL;
    m->last_poses = NULL;
    m->n_last = 0;
    m->last_on_device = 0;
}

/* frames and weights the device tail reads; built on first use */
static int pose_tables(oslam_model *m, oslam_scene *s)
{
    int rc = OSLAM_OK;
    float *h = NULL, *d_T = NULL, *d_w = NULL, *d_S = NULL;
    if (!m->d_Tm16) {
        const size_t M = (size_t)m->c.n;
        h = (float *)malloc(sizeof(float) * 16 * M);
        if (!h) return fail(OSLAM_E_NOMEM, "host allocation failed");
        oslam_T_g_full(m->c.h_xyz, m->c.h_nrm, 0, 1, M, h);
        HIPCHK(hipMalloc((void **)&d_T, sizeof(float) * 16 * M));
        HIPCHK(hipMemcpy(d_T, h, sizeof(float) * 16 * M, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void **)&d_w, sizeof(float) * M));
        HIPCHK(hipMemcpy(d_w, m->weights, sizeof(float) * M, hipMemcpyHostToDevice));
        m->d_Tm16 = d_T;                 /* the model owns them only when both are complete */
        m->d_weights = d_w;
        d_T = d_w = NULL;
        free(h);
        h = NULL;
    }
    if (!s->d_Ts16) {
        const size_t n_all = ((size_t)s->c.n + s->df - 1) / s->df;
        h = (float *)malloc(sizeof(float) * 16 * n_all);
        if (!h) return fail(OSLAM_E_NOMEM, "host allocation failed");
        oslam_T_g_full(s->c.h_xyz, s->c.h_nrm, 0, s->df, n_all, h);
        HIPCHK((hipError_t)oslam_dev_alloc((void **)&d_S, sizeof(float) * 16 * n_all));
        HIPCHK(hipMemcpy(d_S, h, sizeof(float) * 16 * n_all, hipMemcpyHostToDevice));
        s->d_Ts16 = d_S;
        d_S = NULL;
    }
done:
    free(h);
    if (d_T) (void)hipFree(d_T);
    if (d_w) (void)hipFree(d_w);
    oslam_dev_free(d_S);
    return rc;
}

/* the 64 rotations about x of the pose tail (host libm), made once per process */
static float g_rotx[128];
static pthread_once_t g_rotx_once = PTHREAD_ONCE_INIT;
static void rotx_init(void) { oslam_rotx_table(g_rotx); }

/* device buffers for the kept cells and their poses of up to n records */
static int ensure_pose_buffers(oslam_model *m, size_t n)
{
    int rc = OSLAM_OK;
    if (m->pose_cap < n) {
        /* with head room: the number of peak records changes from frame to frame, and freeing device memory waits
         * for the device (0.2 ms a time on the 50-model depth stream) */
        const size_t cap = n + n / 2 > 8192 ? n + n / 2 : 8192;
        if (m->d_pose_cells) (void)hipFree(m->d_pose_cells);
        if (m->d_pose_T) (void)hipFree(m->d_pose_T);
        m->d_pose_cells = NULL;
        m->d_pose_T = NULL;
        m->pose_cap = 0;
        HIPCHK(hipMalloc((void **)&m->d_pose_cells, sizeof(oslamk_cell) * cap));
        HIPCHK(hipMalloc((void **)&m->d_pose_T, sizeof(float) * 16 * cap));
        m->pose_cap = cap;
    }
done:
    return rc;
}

/* Pose tail on the device over the n records in m->d_out.  Returns OSLAM_OK with *done = 1 when it
 * produced the pose; *done = 0 when fewer than two cells survive (the host path handles those). */
static int finish_on_device(oslam_model *m, oslam_scene *s, size_t n, uint32_t gmax, float T[16], oslam_stats *st,
                            int *done)
{
    int rc = OSLAM_OK, k;
    const float *rot = g_rotx;
    uint32_t n_kept = 0, best = 0;
    const float min_votecount = m->params.vote_count_threshold * gmax;      /* model.cu:164 */
    *done = 0;
    rc = pose_tables(m, s);
    if (rc != OSLAM_OK) return rc;
    pthread_once(&g_rotx_once, rotx_init);
    rc = ensure_pose_buffers(m, n);
    if (rc != OSLAM_OK) return rc;
    k = oslamk_pose_stage(m->d_out, (uint32_t)n, min_votecount, m->d_Tm16, s->d_Ts16, s->df, m->d_weights, rot, m->d_dist,
                          m->params.use_l1_norm, m->d_pose_cells, m->d_pose_T, gmax, (uint32_t)m->c.n, (uint32_t)s->c.n, m->params.pose_two_sorts, &n_kept,
                          &best, T, g_stream);
    if (k == -2) return fail(OSLAM_E_NOMEM, "host allocation failed");
    if (k != 0) return fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k));
    if (n_kept < 2) return OSLAM_OK;
    drop_last(m);
    m->n_last = n_kept;
    m->last_on_device = 1;
    if (st) { st->num_top = n_kept; st->max_count = gmax; }
    *done = 1;
    return rc;
}

/* the taps read the last result from the host: fetch it if it is still on the device */
static int materialise_last(oslam_model *m)
{
    int rc = OSLAM_OK;
    const size_t n = m->n_last;
    if (!m->last_on_device) return OSLAM_OK;
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    m->last_cells = (oslam_cell *)malloc(sizeof(oslam_cell) * (n ? n : 1));
    m->last_poses = (float *)malloc(sizeof(float) * 16 * (n ? n : 1));
    if (!m->last_cells || !m->last_poses) { drop_last(m); return fail(OSLAM_E_NOMEM, "host allocation failed"); }
    HIPCHK(hipMemcpy(m->last_cells, m->d_pose_cells, sizeof(oslam_cell) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(m->last_poses, m->d_pose_T, sizeof(float) * 16 * n, hipMemcpyDeviceToHost));
    m->last_on_device = 0;
done:
    return rc;
}

static int finish_cells(oslam_model *m, oslam_scene *s, oslam_cell *cells, size_t n, uint32_t gmax,
                        float T[16], oslam_stats *st)
{
    int rc;
    n = oslam_filter_cells(cells, n, m->params.vote_count_threshold, gmax);
    oslam_sort_cells(cells, n);
    drop_last(m);
    m->last_cells = (oslam_cell *)malloc(sizeof(oslam_cell) * (n ? n : 1));
    m->last_poses = (float *)calloc(16 * (n ? n : 1), sizeof(float));
    m->n_last = 0;
    if (!m->last_cells || !m->last_poses) return fail(OSLAM_E_NOMEM, "host allocation failed");
    memcpy(m->last_cells, cells, sizeof(oslam_cell) * n);
    m->n_last = n;
    if (st) { st->num_top = n; st->max_count = gmax; }
    oslam_pose_set_cluster_hook(cluster_scores_on_device);
    rc = oslam_pose_stage(cells, n, m->c.h_xyz, m->c.h_nrm, (size_t)m->c.n, s->c.h_xyz, s->c.h_nrm,
                          (size_t)s->c.n, m->d_dist, m->params.cpu_clustering, m->params.use_l1_norm,
                          m->params.use_averaged_clusters, m->weights, T, m->last_poses);
    oslam_pose_set_cluster_hook(NULL);
    if (rc == OSLAM_E_NO_VOTES) return fail(rc, "no scene pair matched the model");
    if (rc != OSLAM_OK) return fail(rc, "pose stage failed");
    return OSLAM_OK;
}

int oslam_align_prepare(oslam_model *m, oslam_scene *s)
{
    int rc;
    scratch_pool *pool;
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    pool = pool_lock(m->dev);
    if (!pool) return fail(OSLAM_E_LIMIT, "device ordinal too large");
    rc = pool_reserve_counts(pool, (size_t)(s->n_ref > 0 ? s->n_ref : 1));
    if (rc == OSLAM_OK && pose_gpu_from(m)) rc = pose_tables(m, s);
    pool_unlock(pool);
    return rc;
}

/* everything after the votes of one registration on one device: n records in m->d_out (and in m->h_out
 * unless they were left on the device) */
static int finish_after_votes(oslam_model *m, oslam_scene *s, size_t n, uint32_t gmax, float T[16], oslam_stats *stats)
{
    int rc = OSLAM_OK;
    if (pose_gpu_from(m) && n >= pose_gpu_from(m)) {
        int done = 0;
        rc = finish_on_device(m, s, n, gmax, T, stats, &done);
        if (rc != OSLAM_OK || done) return rc;
        HIPCHK(hipMemcpy(m->h_out, m->d_out, sizeof(oslam_cell) * n, hipMemcpyDeviceToHost));
    }
    rc = finish_cells(m, s, m->h_out, n, gmax, T, stats);
done:
    return rc;
}

int oslam_align(oslam_model *m, oslam_scene *s, float T[16], oslam_stats *stats)
{
    int rc;
    oslamk_counters cnt;
    size_t n = 0;
    oslam_stats local;
    scratch_pool *pool;
    double t0 = now_ms();
    if (!T) return fail(OSLAM_E_INVALID, "T is NULL");
    memset(T, 0, 16 * sizeof(float));
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    pool = pool_lock(m->dev);
    if (!pool) return fail(OSLAM_E_LIMIT, "device ordinal too large");
    g_cur_pool = pool;
    rc = vote_and_fetch(pool, m, s, &cnt, &n, stats, pose_gpu_from(m));
    if (rc == OSLAM_OK) rc = finish_after_votes(m, s, n, cnt.gmax, T, stats);
    g_cur_pool = NULL;
    pool_unlock(pool);
    stats->ms_total = (float)(now_ms() - t0);
    return rc;
}

/* ---- multi-GPU, host-buffer form ------------------------------------------------------------- */
int oslam_align_local(oslam_model *m, oslam_scene *s, oslam_cell *cells_out, size_t cap,
                      size_t *n_out, uint32_t *local_max_out, oslam_stats *stats)
{
    int rc;
    oslamk_counters cnt;
    size_t n = 0;
    oslam_stats local;
    scratch_pool *pool;
    double t0 = now_ms();
    if (!n_out || !local_max_out || (!cells_out && cap)) return fail(OSLAM_E_INVALID, "NULL output");
    *n_out = 0;
    *local_max_out = 0;
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    pool = pool_lock(m->dev);
    if (!pool) return fail(OSLAM_E_LIMIT, "device ordinal too large");
    m->n_local = 0;
    rc = vote_and_fetch(pool, m, s, &cnt, &n, stats, 0);
    pool_unlock(pool);
    if (rc != OSLAM_OK) return rc;
    /* peaks above the local threshold: a superset of what survives the global one; they stay with the
     * model (oslam_local_peaks hands them out again, filtered with the global maximum) */
    n = oslam_filter_cells(m->h_out, n, m->params.vote_count_threshold, cnt.gmax);
    m->n_local = n;
    m->local_max = cnt.gmax;
    *n_out = n;
    *local_max_out = cnt.gmax;
    stats->ms_total = (float)(now_ms() - t0);
    if (n > cap) {
        /* nothing is dropped silently: the caller learns the count and either passes a larger buffer to
         * oslam_local_peaks or exchanges the maxima first and asks for the (fewer) survivors */
        if (cap) {
            oslam_sort_cells(m->h_out, n);
            memcpy(cells_out, m->h_out, sizeof(oslam_cell) * cap);
        }
        return cap ? fail(OSLAM_E_LIMIT, "more local peaks than the buffer holds: *n_out is the number; fetch them with oslam_local_peaks")
                   : OSLAM_OK;
    }
    memcpy(cells_out, m->h_out, sizeof(oslam_cell) * n);
    return OSLAM_OK;
}

int oslam_local_peaks(oslam_model *m, uint32_t global_max, oslam_cell *cells_out, size_t cap, size_t *n_out)
{
    size_t i, n = 0;
    float bound;
    if (!m || !n_out || (!cells_out && cap)) return fail(OSLAM_E_INVALID, "NULL argument");
    if (global_max < m->local_max) return fail(OSLAM_E_INVALID, "the global maximum is below this rank's own");
    bound = m->params.vote_count_threshold * (float)global_max;      /* model.cu:164 */
    for (i = 0; i < m->n_local; i++)
        if ((float)m->h_out[i].count > bound) {
            if (n < cap) cells_out[n] = m->h_out[i];
            n++;
        }
    *n_out = n;
    if (n > cap) return fail(OSLAM_E_LIMIT, "more peaks above the global threshold than the buffer holds: *n_out is the number");
    return OSLAM_OK;
}

int oslam_align_finish(oslam_model *m, oslam_scene *s, const oslam_cell *cells, size_t n,
                       uint32_t global_max, float T[16], oslam_stats *stats)
{
    int rc;
    oslam_cell *tmp = NULL;
    oslam_stats local;
    scratch_pool *pool;
    if (!T || (!cells && n)) return fail(OSLAM_E_INVALID, "NULL argument");
    memset(T, 0, 16 * sizeof(float));
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    pool = pool_lock(m->dev);
    if (!pool) return fail(OSLAM_E_LIMIT, "device ordinal too large");
    g_cur_pool = pool;
    if (pose_gpu_from(m) && n >= pose_gpu_from(m) && n <= m->out_cap) {
        /* the gathered union goes back to HBM; codes that do not name a reference point of this scene
         * and a point of this model are left to the host path, which reports them */
        size_t i;
        int ok = 1, done = 0;
        for (i = 0; i < n && ok; i++) {
            const uint32_t sr = (uint32_t)(cells[i].code >> 32), mr = ((uint32_t)cells[i].code) >> 6;
            ok = sr < (uint32_t)s->c.n && sr % s->df == 0 && mr < (uint32_t)m->c.n;
        }
        if (ok) {
            HIPCHK(hipMemcpy(m->d_out, cells, sizeof(oslam_cell) * n, hipMemcpyHostToDevice));
            rc = finish_on_device(m, s, n, global_max, T, stats, &done);
            if (rc != OSLAM_OK || done) goto done;
        }
    }
    tmp = (oslam_cell *)malloc(sizeof(oslam_cell) * (n ? n : 1));
    if (!tmp) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    memcpy(tmp, cells, sizeof(oslam_cell) * n);
    rc = finish_cells(m, s, tmp, n, global_max, T, stats);
done:
    free(tmp);
    g_cur_pool = NULL;
    pool_unlock(pool);
    return rc;
}

/* ---- multi-GPU: one call per rank does everything (ppf.h:9-15 is one call too) -----------------
 * The exchange stays in HBM: all-reduce(MAX) of the vote maxima, the local records filtered with the
 * global threshold where they lie, an all-gather of the survivor counts, an all-gather with exact
 * sizes of the survivors straight into the union buffer, and the pose tail on the union -- on the
 * device when it is large.  The collectives go through the communicator's table of operations
 * (oslam_comm.h): RCCL over xGMI, or the in-process loopback that lets the same function run with
 * N emulated ranks on one device.  Latency-bound: a few KiB to a few hundred KiB per rank.
 *
 * Failure is collective: whatever goes wrong on ONE rank between two collectives (no memory for the
 * union, more peaks than the buffers can hold, a failed kernel) travels as an error word beside the
 * payload of the next collective, so that every rank leaves at the same point -- none is left
 * waiting in a collective its peer will never enter.  A collective that fails itself aborts the
 * communicator (ncclCommAbort) and marks it broken. */
static int peer_failed(void)
{
    return fail(OSLAM_E_PEER, "a peer rank failed: the registration was abandoned on every rank");
}

/* this rank's n_local records in m->d_out (local maximum lmax, rc_local = what the vote stage returned) ->
 * the union of every rank's records above the global threshold in m->d_out, *total of them */
static int exchange_peaks(oslam_model *m, oslam_comm *c, size_t n_local, uint32_t lmax, int rc_local,
                          uint32_t *gmax_out, size_t *total_out)
{
    int rc = OSLAM_OK, r, any = 0, grow_any = 0, together = 0;   /* together: every rank leaves at this point */
    uint32_t n_mine = 0, err, gmax;
    size_t total = 0, bytes[64];
    size_t *by = bytes;
    hipStream_t st = (hipStream_t)g_stream;
    uint32_t *h = c->h_small, *d = c->d_small;
    *gmax_out = 0;
    *total_out = 0;
    if (c->world > 64) {
        by = (size_t *)malloc(sizeof(size_t) * (size_t)c->world);
        if (!by) { by = bytes; rc_local = rc_local != OSLAM_OK ? rc_local : fail(OSLAM_E_NOMEM, "host allocation failed"); }
    }
    /* 1. the threshold is global (model.cu:164-170): maximum over ranks, with the error word */
    h[0] = lmax;
    h[1] = (rc_local != OSLAM_OK || c->inject_stage == OSLAM_STAGE_VOTE) ? 1u : 0u;
    HIPCHK(hipMemcpyAsync(d, h, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    rc = oslam_comm_all_reduce_max(c, d, 2, g_stream);
    if (rc != OSLAM_OK) goto done;
    HIPCHK(hipMemcpyAsync(h, d, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h[1]) {
        together = 1;
        rc = rc_local != OSLAM_OK ? rc_local : c->inject_stage == OSLAM_STAGE_VOTE ? fail(OSLAM_E_DEVICE, "injected failure (vote stage)") : peer_failed();
        goto done;
    }
    gmax = h[0];
    /* 2. survivors of this rank, compacted into the second record buffer */
    err = 0;
    {
        scratch_pool *pool = pool_lock(m->dev);          /* the selection shares the device's work space */
        if (!pool || ensure_union(m, n_local > 0 ? n_local : 1) != OSLAM_OK) err = 1;
        else if (n_local) {
            const int k = oslamk_select_cells(m->d_out, (uint32_t)n_local, m->params.vote_count_threshold * (float)gmax, m->d_union,
                                              &n_mine, g_stream);
            if (k != 0) { (void)fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)k)); err = 1; }
        }
        pool_unlock(pool);
    }
    if (c->inject_stage == OSLAM_STAGE_SELECT) err = 1;
    h[0] = n_mine;
    h[1] = err;
    h[2] = m->out_cap;
    HIPCHK(hipMemcpyAsync(d, h, 3 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    rc = oslam_comm_all_gather(c, d, d + 4, 3, g_stream);
    if (rc != OSLAM_OK) goto done;
    HIPCHK(hipMemcpyAsync(h + 4, d + 4, 3 * sizeof(uint32_t) * (size_t)c->world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (r = 0; r < c->world; r++) {
        total += h[4 + 3 * r];
        any |= h[4 + 3 * r + 1] != 0;
        by[r] = (size_t)h[4 + 3 * r] * sizeof(oslamk_cell);
    }
    if (any) {
        together = 1;
        rc = err ? (c->inject_stage == OSLAM_STAGE_SELECT ? fail(OSLAM_E_DEVICE, "injected failure (selection stage)")
                                                           : fail(OSLAM_E_NOMEM, "no device memory for this rank's survivors"))
                 : peer_failed();
        goto done;
    }
    if (total > ((size_t)1 << 28)) {                       /* the same on every rank */
        together = 1;
        rc = fail(OSLAM_E_LIMIT, "more than 2^28 accumulator peaks above the threshold");
        goto done;
    }
    /* 3. room for the union, rank after rank, in m->d_out (this rank's survivors are safe in d_union).  Buffers
     * differ per rank; whether ANY rank has to grow is known to all from the gathered capacities, and only
     * then does everybody meet once more to learn whether the growing worked */
    for (r = 0; r < c->world; r++) grow_any |= total > h[4 + 3 * r + 2];
    if (grow_any) {
        err = 0;
        if (total > m->out_cap && grow_records(m, total + total / 8 + 1024) != OSLAM_OK) err = 1;
        if (c->inject_stage == OSLAM_STAGE_GROW) err = 1;
        h[0] = err;
        HIPCHK(hipMemcpyAsync(d, h, sizeof(uint32_t), hipMemcpyHostToDevice, st));
        rc = oslam_comm_all_reduce_max(c, d, 1, g_stream);
        if (rc != OSLAM_OK) goto done;
        HIPCHK(hipMemcpyAsync(h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h[0]) {
            together = 1;
            rc = err ? (c->inject_stage == OSLAM_STAGE_GROW ? fail(OSLAM_E_DEVICE, "injected failure (growing the record buffer)")
                                                             : fail(OSLAM_E_NOMEM, "no memory for the union of the accumulator peaks"))
                     : peer_failed();
            goto done;
        }
    }
    /* 4. the union */
    if (total) {
        rc = oslam_comm_all_gather_v(c, m->d_union, m->d_out, by, g_stream);
        if (rc != OSLAM_OK) goto done;
        HIPCHK(hipStreamSynchronize(st));
    }
    *gmax_out = gmax;
    *total_out = total;
done:
    if (rc != OSLAM_OK && !together && !c->broken) {
        /* a HIP call of this rank failed between collectives: the peers cannot be told through a device buffer any
         * more; give the communicator up so that nothing of it is used again */
        c->ops->abort(c->ctx, c->rank);
        c->broken = 1;
    }
    c->inject_stage = OSLAM_STAGE_NONE;
    if (by != bytes) free(by);
    return rc;
}

int oslam_align_multi(oslam_model *m, oslam_scene *s, oslam_comm *c, float T[16], oslam_stats *stats)
{
    int rc, vrc;
    oslamk_counters cnt;
    size_t n = 0, total = 0;
    uint32_t gmax = 0;
    oslam_stats local;
    scratch_pool *pool = NULL;
    double t0 = now_ms();
    if (!T || !c) return fail(OSLAM_E_INVALID, "NULL argument");
    memset(T, 0, 16 * sizeof(float));
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (c->dev != m->dev) return fail(OSLAM_E_INVALID, "communicator and model live on different devices");
    if (s->world != c->world || s->rank != c->rank) return fail(OSLAM_E_INVALID, "the scene's shard differs from the communicator's rank");
    if (c->broken) return fail(OSLAM_E_DEVICE, "the communicator was aborted after a failed collective: make a new one");
    if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    /* this rank's votes; the records stay in m->d_out.  The device's pool is held for the votes and for the pose
     * tail, not across the collectives: emulated ranks share a device (and a pool) */
    memset(&cnt, 0, sizeof cnt);
    pool = pool_lock(m->dev);
    if (!pool) vrc = fail(OSLAM_E_LIMIT, "device ordinal too large");
    else {
        g_cur_pool = pool;
        vrc = vote_and_fetch(pool, m, s, &cnt, &n, stats, 1);
        g_cur_pool = NULL;
        pool_unlock(pool);
    }
    if (vrc != OSLAM_OK) { n = 0; cnt.gmax = 0; }
    rc = exchange_peaks(m, c, n, cnt.gmax, vrc, &gmax, &total);
    if (rc != OSLAM_OK) goto done;
    stats->num_emitted = (uint32_t)total;
    /* every rank finishes on the same union: same pose everywhere, no second exchange */
    pool = pool_lock(m->dev);
    g_cur_pool = pool;
    if (pose_gpu_from(m) && total >= pose_gpu_from(m)) {
        int done = 0;
        rc = finish_on_device(m, s, total, gmax, T, stats, &done);
        if (rc != OSLAM_OK || done) goto unlock;
    }
    if (total && hipMemcpy(m->h_out, m->d_out, sizeof(oslam_cell) * total, hipMemcpyDeviceToHost) != hipSuccess) {
        rc = fail(OSLAM_E_DEVICE, "hipMemcpy of the union failed");
        goto unlock;
    }
    rc = finish_cells(m, s, m->h_out, total, gmax, T, stats);
unlock:
    g_cur_pool = NULL;
    pool_unlock(pool);
done:
    stats->ms_total = (float)(now_ms() - t0);
    return rc;
}

/* ------------------------------------------------------------------------
 * Model database (src/cuda/ppf.cu:57-100: the reference loops scenes x models and rebuilds
 * both every time).  Models that share d_dist (and device and vote mode) form a group with ONE union table:
 * the scene pass -- count, pair keys, probe, hit sort -- then runs once per group and frame instead of once
 * per model, every model votes from the same hit lists with its own buckets (table.uinfo under the group's
 * slots), and nothing waits on the host between the models of a group.  Models with a d_dist of their own
 * are groups of one and take the single-model path.
 * ---------------------------------------------------------------------- */
typedef struct db_group {
    int n;
    size_t *members;                  /* indices into db->models */
    uint32_t *ukeys, *reach, *kmap, *uids;   /* the group's union table, reachable-distance bitset, key map and key numbers (n > 1) */
} db_group;

struct oslam_db {
    int dev;
    size_t n;
    oslam_model **models;             /* borrowed */
    int n_groups;
    db_group *groups;
};

static int same_group(const oslam_model *a, const oslam_model *b)
{
    return a->dev == b->dev && a->d_dist == b->d_dist && a->params.vote_mode == b->params.vote_mode;
}

static void db_destroy(oslam_db *db, int give_back)
{
    int g, k;
    uint32_t *d_small = NULL;
    if (!db) return;
    (void)hipSetDevice(db->dev);
    if (give_back) (void)hipMalloc((void **)&d_small, 2 * sizeof(uint32_t));
    for (g = 0; g < db->n_groups; g++) {
        db_group *gr = &db->groups[g];
        if (gr->n > 1) {
            /* the members get a union table of their own back; one that cannot (or is not asked to) is left
             * without key tables and refuses every call but oslam_model_destroy */
            for (k = 0; k < gr->n; k++) {
                oslam_model *m = db->models[gr->members[k]];
                int ok = 0;
                if (!m->shared_union) continue;         /* never switched to the group's tables (a failed oslam_db_create) */
                if (d_small && build_union(m, (uint32_t)(m->num_model_keys ? m->num_model_keys - 1 : 0), d_small, d_small + 1) == OSLAM_OK)
                    ok = build_uinfo(m) == OSLAM_OK;
                if (!ok) {
                    if (m->shared_union) {               /* still pointing at the group's tables, which are freed below */
                        m->table.ukeys = NULL;
                        m->table.reach = NULL;
                        m->table.kmap = NULL;
                        m->table.uids = NULL;
                        m->shared_union = 0;
                    }
                    m->unusable = 1;
                }
            }
            (void)hipStreamSynchronize((hipStream_t)g_stream);
            if (gr->ukeys) (void)hipFree(gr->ukeys);
            if (gr->reach) (void)hipFree(gr->reach);
            if (gr->kmap) (void)hipFree(gr->kmap);
            if (gr->uids) (void)hipFree(gr->uids);
        }
        free(gr->members);
    }
    if (d_small) (void)hipFree(d_small);
    free(db->groups);
    free(db->models);
    free(db);
}

void oslam_db_destroy(oslam_db *db) { db_destroy(db, 1); }

void oslam_db_destroy_with_models(oslam_db *db) { db_destroy(db, 0); }

int oslam_db_create(oslam_model *const *models, size_t n, oslam_db **out)
{
    int rc = OSLAM_OK, g, k;
    size_t j;
    oslam_db *db;
    uint32_t *d_small = NULL, h_small[2];
    if (!out) return fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (!models || n == 0) return fail(OSLAM_E_INVALID, "empty database");
    for (j = 0; j < n; j++) {
        if (!models[j]) return fail(OSLAM_E_INVALID, "NULL model");
        if (models[j]->shared_union) return fail(OSLAM_E_INVALID, "a model can be in one database at a time");
        if (models[j]->unusable) return fail(OSLAM_E_INVALID, "a model without key tables cannot join a database");
        if (models[j]->dev != models[0]->dev) return fail(OSLAM_E_INVALID, "the models of a database live on one device");
        for (k = 0; k < (int)j; k++)
            if (models[k] == models[j]) return fail(OSLAM_E_INVALID, "the same model handle twice in one database");
    }
    db = (oslam_db *)calloc(1, sizeof *db);
    if (!db) return fail(OSLAM_E_NOMEM, "host allocation failed");
    db->dev = models[0]->dev;
    db->n = n;
    db->models = (oslam_model **)malloc(sizeof *db->models * n);
    db->groups = (db_group *)calloc(n, sizeof *db->groups);
    if (!db->models || !db->groups) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    memcpy(db->models, models, sizeof *db->models * n);
    if (hipSetDevice(db->dev) != hipSuccess) { rc = fail(OSLAM_E_DEVICE, "hipSetDevice failed"); goto done; }
    for (j = 0; j < n; j++) {
        for (g = 0; g < db->n_groups; g++)
            if (same_group(models[db->groups[g].members[0]], models[j])) break;
        if (g == db->n_groups) {
            db->groups[g].members = (size_t *)malloc(sizeof(size_t) * n);
            if (!db->groups[g].members) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
            db->n_groups++;
        }
        db->groups[g].members[db->groups[g].n++] = j;
    }
    HIPCHK(hipMalloc((void **)&d_small, 2 * sizeof(uint32_t)));
    for (g = 0; g < db->n_groups; g++) {
        db_group *gr = &db->groups[g];
        oslamk_table t;
        uint64_t distinct = 0;
        uint32_t lg = 16;
        if (gr->n < 2) continue;
        for (k = 0; k < gr->n; k++) distinct += models[gr->members[k]]->num_model_keys;
        while (((uint64_t)1 << lg) < 4u * distinct && lg < OSLAMK_RUN_SHIFT) lg++;
        if (((uint64_t)1 << lg) < 2u * distinct) { rc = fail(OSLAM_E_LIMIT, "more distinct pair keys in the group than a union table can index"); goto done; }
        HIPCHK(hipMalloc((void **)&gr->ukeys, sizeof(uint32_t) << lg));
        HIPCHK(hipMalloc((void **)&gr->reach, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32)));
        HIPCHK(hipMemsetAsync(gr->ukeys, 0, sizeof(uint32_t) << lg, (hipStream_t)g_stream));
        HIPCHK(hipMemsetAsync(gr->reach, 0, sizeof(uint32_t) * (OSLAMK_REACH_BINS / 32), (hipStream_t)g_stream));
        HIPCHK(hipMemsetAsync(d_small, 0, 2 * sizeof(uint32_t), (hipStream_t)g_stream));
        /* every member's keys into the group's table */
        for (k = 0; k < gr->n; k++) {
            t = models[gr->members[k]]->table;
            t.ukeys = gr->ukeys;
            t.ucap = 1u << lg;
            t.ushift = 32 - lg;
            KCHK(oslamk_union_build(t, d_small, d_small + 1, g_stream));
        }
        t.reach = gr->reach;
        KCHK(oslamk_reach_build(t, models[gr->members[0]]->d_dist, g_stream));
        rc = build_kmap(&t, models[gr->members[0]]->d_dist);
        gr->kmap = t.kmap;
        gr->uids = t.uids;
        if (rc != OSLAM_OK) goto done;
        HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
        HIPCHK(hipMemcpy(h_small, d_small, sizeof h_small, hipMemcpyDeviceToHost));
        if (h_small[1]) { rc = fail(OSLAM_E_LIMIT, "union key table overflow"); goto done; }
        /* the members look their buckets up under the group's slots from now on */
        for (k = 0; k < gr->n; k++) {
            oslam_model *m = models[gr->members[k]];
            (void)hipFree(m->table.ukeys);
            (void)hipFree(m->table.reach);
            if (m->table.kmap) (void)hipFree(m->table.kmap);
            if (m->table.uids) (void)hipFree(m->table.uids);
            m->table.ukeys = gr->ukeys;
            m->table.reach = gr->reach;
            m->table.kmap = gr->kmap;
            m->table.uids = gr->uids;
            m->table.kmap_bins = t.kmap_bins;
            m->table.reach_words = t.reach_words;
            m->table.n_ids = t.n_ids;
            m->table.id_bits = t.id_bits;
            m->table.uinfo_stride = t.uinfo_stride;
            m->table.ucap = 1u << lg;
            m->table.ushift = 32 - lg;
            m->shared_union = 1;
            rc = build_uinfo(m);
            if (rc != OSLAM_OK) goto done;
        }
        HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    }
done:
    if (d_small) (void)hipFree(d_small);
    if (rc != OSLAM_OK) { oslam_db_destroy(db); return rc; }
    *out = db;
    return OSLAM_OK;
}

int oslam_db_align(oslam_db *db, oslam_scene *s, float *T_out, oslam_stats *stats)
{
    int rc = OSLAM_OK, g, k, first_err = OSLAM_OK;
    scratch_pool *pool;
    oslamk_counters *cnt = NULL;
    oslam_model **ms = NULL;
    int *on_dev = NULL;               /* per member of the current group: 0, or the number of cells its device tail kept */
    double t0 = now_ms();
    if (!db || !s || !T_out) return fail(OSLAM_E_INVALID, "NULL argument");
    memset(T_out, 0, sizeof(float) * 16 * db->n);
    if (stats) memset(stats, 0, sizeof *stats * db->n);
    for (k = 0; k < (int)db->n; k++) {
        rc = check_pair(db->models[k], s);
        if (rc != OSLAM_OK) return rc;
    }
    if (hipSetDevice(db->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    cnt = (oslamk_counters *)malloc(sizeof *cnt * db->n);
    ms = (oslam_model **)malloc(sizeof *ms * db->n);
    on_dev = (int *)malloc(sizeof *on_dev * (db->n ? db->n : 1));
    if (!cnt || !ms || !on_dev) { free(cnt); free(ms); free(on_dev); return fail(OSLAM_E_NOMEM, "host allocation failed"); }
    pool = pool_lock(db->dev);
    if (!pool) { free(cnt); free(ms); free(on_dev); return fail(OSLAM_E_LIMIT, "device ordinal too large"); }
    g_cur_pool = pool;
    for (g = 0; g < db->n_groups && rc == OSLAM_OK; g++) {
        db_group *gr = &db->groups[g];
        float ms_all = 0.0f, msv = 0.0f, msk = 0.0f;
        uint32_t launches = 0;
        uint64_t probed = 0;
        for (k = 0; k < gr->n; k++) ms[k] = db->models[gr->members[k]];
        if (gr->n > 1) {
            /* one scene pass, every member's votes behind it */
            rc = run_votes_group(pool, ms, gr->n, s, s->d_ref_idx, s->d_tsg, s->n_ref, 0, NULL, cnt, &ms_all, &msv, &msk,
                                 &launches, &probed);
            if (rc != OSLAM_OK) break;
        }
        /* The pose tails of the group's members, in flight together: every member's selection of its peak records is
         * enqueued, one wait, then every member's chain (order, poses, clustering scores, winner), one wait -- two
         * waits per group instead of two per model, and the kernels of one model run while the next one's are being
         * launched (50 models on a depth frame: 36 -> 32.6 ms together with the single packed sort).  A member whose records did not fit its buffer, whose
         * tail belongs to the host (few records, or a host-only variant), or with fewer than two records above the
         * threshold goes through the single-model path afterwards. */
        memset(on_dev, 0, sizeof *on_dev * (size_t)gr->n);
        if (gr->n > 1) {
            uint32_t n_max = 0;
            for (k = 0; k < gr->n; k++) {
                oslam_model *m = ms[k];
                if (cnt[k].out_count <= m->out_cap && pose_gpu_from(m) && cnt[k].out_count >= pose_gpu_from(m)) {
                    on_dev[k] = 1;
                    if (cnt[k].out_count > n_max) n_max = (uint32_t)cnt[k].out_count;
                }
            }
            if (n_max) {
                int kk;
                pthread_once(&g_rotx_once, rotx_init);
                for (k = 0; k < gr->n && rc == OSLAM_OK; k++)
                    if (on_dev[k]) {
                        rc = pose_tables(ms[k], s);
                        if (rc == OSLAM_OK) rc = ensure_pose_buffers(ms[k], (size_t)cnt[k].out_count);
                    }
                if (rc != OSLAM_OK) break;
                kk = oslamk_pose_reserve(n_max, (uint32_t)gr->n, g_rotx, g_stream);
                for (k = 0; k < gr->n && kk == 0; k++)
                    if (on_dev[k])
                        kk = oslamk_pose_select_async(ms[k]->d_out, (uint32_t)cnt[k].out_count,
                                                      ms[k]->params.vote_count_threshold * cnt[k].gmax, ms[k]->d_pose_cells,
                                                      (uint32_t)k, g_stream);
                if (kk == 0) kk = (int)hipStreamSynchronize((hipStream_t)g_stream);
                for (k = 0; k < gr->n && kk == 0; k++)
                    if (on_dev[k]) {
                        const uint32_t n_sel = oslamk_pose_selected((uint32_t)k);
                        if (n_sel < 2) { on_dev[k] = 0; continue; }
                        on_dev[k] = (int)n_sel;
                        kk = oslamk_pose_finish_async(n_sel, ms[k]->d_pose_cells, ms[k]->d_Tm16, s->d_Ts16, s->df, ms[k]->d_weights,
                                                      ms[k]->d_dist, ms[k]->params.use_l1_norm, ms[k]->d_pose_cells,
                                                      ms[k]->d_pose_T, cnt[k].gmax, (uint32_t)ms[k]->c.n, (uint32_t)s->c.n,
                                                      ms[k]->params.pose_two_sorts, (uint32_t)k, g_stream);
                    }
                if (kk == 0) kk = (int)hipStreamSynchronize((hipStream_t)g_stream);
                if (kk != 0) { rc = fail(OSLAM_E_DEVICE, hipGetErrorString((hipError_t)kk)); break; }
            }
        }
        for (k = 0; k < gr->n && rc == OSLAM_OK; k++) {
            oslam_model *m = ms[k];
            oslam_stats local, *st = stats ? &stats[gr->members[k]] : &local;
            float *T = T_out + 16 * gr->members[k];
            size_t n = 0;
            int arc;
            memset(st, 0, sizeof *st);
            if (gr->n > 1 && cnt[k].out_count <= m->out_cap) {
                n = cnt[k].out_count;
                st->num_scene_ppfs = (uint64_t)s->n_ref * (uint64_t)(s->c.n - 1);
                st->num_hits = cnt[k].hits;                /* of the group's pass: pairs whose key is in some member */
                st->num_votes = cnt[k].votes;
                st->num_unique_votes = cnt[k].nonzero_cells;
                st->num_model_keys = m->num_model_keys;
                st->max_count = cnt[k].gmax;
                st->num_emitted = cnt[k].out_count;
                st->ms_vote = ms_all / (float)gr->n;       /* the group's kernels, shared out evenly */
                st->ms_vote_kernel = msv / (float)gr->n;
                st->ms_key_kernel = msk / (float)gr->n;
                st->vote_launches = launches;
                st->num_pairs_probed = probed;
                st->scratch_bytes = pool->bytes;
                st->num_entries_streamed = cnt[k].entries;
                st->num_items = cnt[k].items;
                st->wide_workgroups = cnt[k].redo_total;
                if (on_dev[k]) {                            /* its chain has run: the winner is in its slot */
                    uint32_t best = 0;
                    oslamk_pose_result((uint32_t)k, &best, T);
                    drop_last(m);
                    m->n_last = (size_t)on_dev[k];
                    m->last_on_device = 1;
                    st->num_top = (uint64_t)on_dev[k];
                    st->ms_total = (float)(now_ms() - t0);
                    continue;
                }
                if (n) HIPCHK(hipMemcpy(m->h_out, m->d_out, sizeof(oslam_cell) * n, hipMemcpyDeviceToHost));
                arc = finish_cells(m, s, m->h_out, n, cnt[k].gmax, T, st);
            } else {
                /* a group of one -- or a member whose peak records did not fit its buffer: the single-model path */
                arc = vote_and_fetch(pool, m, s, &cnt[k], &n, st, pose_gpu_from(m));
                if (arc == OSLAM_OK) arc = finish_after_votes(m, s, n, cnt[k].gmax, T, st);
            }
            if (arc != OSLAM_OK && arc != OSLAM_E_NO_VOTES) rc = arc;
            else if (arc == OSLAM_E_NO_VOTES && first_err == OSLAM_OK) first_err = arc;
            st->ms_total = (float)(now_ms() - t0);
        }
    }
done:
    g_cur_pool = NULL;
    pool_unlock(pool);
    free(cnt);
    free(ms);
    free(on_dev);
    (void)first_err;                  /* a model without votes leaves its T zero, as oslam_ppf_registration does */
    return rc;
}

int oslam_db_size(const oslam_db *db, size_t *n_models, size_t *n_groups)
{
    if (!db) return fail(OSLAM_E_INVALID, "NULL handle");
    if (n_models) *n_models = db->n;
    if (n_groups) *n_groups = (size_t)db->n_groups;
    return OSLAM_OK;
}

/* The database split by model (see oslam.h): this rank's models against the whole scene, then every pose to every
 * rank in one all-gather of {found, error, 16 floats} per model slot.  An error on one rank travels in its slots'
 * error word: every rank returns (OSLAM_E_PEER on the others). */
int oslam_db_align_multi(oslam_db *db, oslam_scene *s, oslam_comm *c, size_t n_total, float *T_out, int *found_out,
                         oslam_stats *stats_local)
{
    int rc = OSLAM_OK, lrc = OSLAM_OK, any = 0, r;
    size_t n_mine, block, k, words;
    uint32_t *h_send = NULL, *h_recv = NULL, *d_buf = NULL;
    float *T_loc = NULL;
    hipStream_t st = (hipStream_t)g_stream;
    if (!s || !c || !T_out || n_total == 0) return fail(OSLAM_E_INVALID, "NULL argument");
    if (c->broken) return fail(OSLAM_E_DEVICE, "the communicator was aborted after a failed collective: make a new one");
    if (s->world != 1) return fail(OSLAM_E_INVALID, "a database split by model takes the whole scene on every rank (shard_world 1)");
    n_mine = (n_total + (size_t)c->world - 1 - (size_t)c->rank) / (size_t)c->world;     /* models rank, rank + world, ... */
    block = (n_total + (size_t)c->world - 1) / (size_t)c->world;
    if ((db ? db->n : 0) != n_mine) return fail(OSLAM_E_INVALID, "this rank's database does not hold models rank, rank + world, ... of n_total");
    memset(T_out, 0, sizeof(float) * 16 * n_total);
    if (found_out) memset(found_out, 0, sizeof(int) * n_total);
    words = 18 * block;
    h_send = (uint32_t *)calloc(words ? words : 1, sizeof(uint32_t));
    h_recv = (uint32_t *)malloc(sizeof(uint32_t) * (words ? words : 1) * (size_t)c->world);
    T_loc = (float *)calloc(16 * (n_mine ? n_mine : 1), sizeof(float));
    if (!h_send || !h_recv || !T_loc) lrc = fail(OSLAM_E_NOMEM, "host allocation failed");
    if (hipSetDevice(c->dev) != hipSuccess) lrc = fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    if (lrc == OSLAM_OK && n_mine) lrc = oslam_db_align(db, s, T_loc, stats_local);
    if (lrc == OSLAM_OK && hipMalloc((void **)&d_buf, sizeof(uint32_t) * words * (size_t)(c->world + 1)) != hipSuccess) {
        d_buf = NULL;
        lrc = fail(OSLAM_E_NOMEM, "no device memory for the pose exchange");
    }
    if (!h_send || !h_recv || !d_buf) {          /* cannot even take part in the collective: the communicator is given up */
        if (!c->broken) { c->ops->abort(c->ctx, c->rank); c->broken = 1; }
        rc = lrc;
        goto done;
    }
    for (k = 0; k < block; k++) {
        uint32_t *slot = h_send + 18 * k;
        slot[1] = lrc != OSLAM_OK;
        if (k < n_mine && lrc == OSLAM_OK) {
            int nz = 0, q;
            for (q = 0; q < 16; q++) nz |= T_loc[16 * k + q] != 0.0f;
            slot[0] = (uint32_t)nz;             /* a model without votes leaves its pose all zeros */
            memcpy(slot + 2, T_loc + 16 * k, 16 * sizeof(float));
        }
    }
    HIPCHK(hipMemcpyAsync(d_buf, h_send, sizeof(uint32_t) * words, hipMemcpyHostToDevice, st));
    rc = oslam_comm_all_gather(c, d_buf, d_buf + words, words, g_stream);
    if (rc != OSLAM_OK) goto done;
    HIPCHK(hipMemcpyAsync(h_recv, d_buf + words, sizeof(uint32_t) * words * (size_t)c->world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (r = 0; r < c->world; r++)
        for (k = 0; k < block; k++) any |= h_recv[((size_t)r * block + k) * 18 + 1] != 0;
    if (any) { rc = lrc != OSLAM_OK ? lrc : peer_failed(); goto done; }
    for (r = 0; r < c->world; r++)
        for (k = 0; k < block; k++) {
            const size_t j = k * (size_t)c->world + (size_t)r;
            const uint32_t *slot = h_recv + ((size_t)r * block + k) * 18;
            if (j >= n_total) continue;
            memcpy(T_out + 16 * j, slot + 2, 16 * sizeof(float));
            if (found_out) found_out[j] = (int)slot[0];
        }
done:
    free(h_send);
    free(h_recv);
    free(T_loc);
    if (d_buf) (void)hipFree(d_buf);
    return rc;
}

int oslam_ppf_registration(const float *const *scene_xyz, const float *const *scene_nrm,
                           const size_t *scene_n, size_t n_scenes, const float *const *model_xyz,
                           const float *const *model_nrm, const size_t *model_n, size_t n_models,
                           size_t stride_bytes, const float *model_d_dists, unsigned df,
                           float vote_count_threshold, int cpu_clustering, int use_l1_norm,
                           int use_averaged_clusters, int devUse, const float *model_weights,
                           float *T_out)
{
    oslam_params p;
    oslam_model **models = NULL;
    oslam_db *db = NULL;
    size_t i, j;
    int rc = OSLAM_OK;
    (void)model_weights;                       /* ignored by the reference too: ppf.cu:35 */
    if (!scene_xyz || !scene_nrm || !scene_n || !model_xyz || !model_nrm || !model_n || !model_d_dists || !T_out)
        return fail(OSLAM_E_INVALID, "NULL argument");
    oslam_params_default(&p);
    p.ref_point_df = df;
    p.vote_count_threshold = vote_count_threshold;
    p.cpu_clustering = cpu_clustering;
    p.use_l1_norm = use_l1_norm;
    p.use_averaged_clusters = use_averaged_clusters;
    p.dev = devUse;
    memset(T_out, 0, sizeof(float) * 16 * n_scenes * n_models);
    models = (oslam_model **)calloc(n_models ? n_models : 1, sizeof *models);
    if (!models) return fail(OSLAM_E_NOMEM, "host allocation failed");
    /* models are built once and stay resident (the reference rebuilds per pair) */
    for (j = 0; j < n_models && rc == OSLAM_OK; j++)
        rc = oslam_model_create(model_xyz[j], model_nrm[j], model_n[j], stride_bytes, model_d_dists[j], &p, &models[j]);
    if (rc == OSLAM_OK && n_models) rc = oslam_db_create(models, n_models, &db);
    for (i = 0; i < n_scenes && rc == OSLAM_OK && n_models; i++) {
        /* one scene object for all models: the reference prepares the scene per model because its
         * pair keys depend on the model's d_dist (ppf.cu:64-67); here they are made inside the align,
         * once per group of models that share a d_dist */
        oslam_scene *sc = NULL;
        rc = oslam_scene_create(scene_xyz[i], scene_nrm[i], scene_n[i], stride_bytes, 0.0f, df, &p, &sc);
        if (rc == OSLAM_OK) rc = oslam_db_align(db, sc, T_out + 16 * (i * n_models), NULL);
        oslam_scene_destroy(sc);
    }
    oslam_db_destroy_with_models(db);         /* the models go next: no key tables are rebuilt for them */
    for (j = 0; j < n_models; j++) oslam_model_destroy(models[j]);
    free(models);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* parity taps */
static int cloud_row_keys(cloud_buf *c, size_t ref, float d_dist, uint32_t *keys_out)
{
    int rc = OSLAM_OK;
    uint32_t *d = NULL;
    if (!keys_out || ref >= (size_t)c->n) return fail(OSLAM_E_INVALID, "bad reference index");
    HIPCHK(hipMalloc((void **)&d, sizeof(uint32_t) * c->n));
    KCHK(oslamk_row_keys(c->k, (int)ref, d_dist, 1.0f / d_dist, d, g_stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    HIPCHK(hipMemcpy(keys_out, d, sizeof(uint32_t) * c->n, hipMemcpyDeviceToHost));
done:
    if (d) (void)hipFree(d);
    return rc;
}

int oslam_scene_keys(oslam_scene *s, size_t ref_index, uint32_t *keys_out)
{
    if (!s) return fail(OSLAM_E_INVALID, "NULL handle");
    if (!(s->d_dist > 0.0f)) return fail(OSLAM_E_INVALID, "this scene was made for models of any d_dist (d_dist 0): it has no keys of its own");
    if (hipSetDevice(s->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    return cloud_row_keys(&s->c, ref_index, s->d_dist, keys_out);
}

int oslam_model_keys(oslam_model *m, size_t ref_index, uint32_t *keys_out)
{
    if (!m) return fail(OSLAM_E_INVALID, "NULL handle");
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    return cloud_row_keys(&m->c, ref_index, m->d_dist, keys_out);
}

static int u32_order(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

int oslam_model_bucket(oslam_model *m, uint32_t key, uint32_t *pairs_out, size_t cap, size_t *count_out)
{
    int rc = OSLAM_OK;
    size_t total = 0, written = 0, n_slots;
    int s;
    uint32_t *tmp = NULL;
    uint16_t *tmi = NULL;
    if (!m || !count_out) return fail(OSLAM_E_INVALID, "NULL argument");
    *count_out = 0;
    if (key == 0) return OSLAM_OK;             /* never matched: kernel.cu:491 */
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    n_slots = (size_t)m->table.cap * m->table.n_slices;
    if (!m->h_slots) {
        m->h_slots = (oslamk_slot *)malloc(sizeof(oslamk_slot) * n_slots);
        if (!m->h_slots) return fail(OSLAM_E_NOMEM, "host allocation failed");
        HIPCHK(hipMemcpy(m->h_slots, m->table.slots, sizeof(oslamk_slot) * n_slots, hipMemcpyDeviceToHost));
    }
    for (s = 0; s < m->table.n_slices; s++) {
        const oslamk_slot *tab = m->h_slots + (size_t)s * m->table.cap;
        uint32_t mask = m->table.cap - 1, slot = (key * 2654435761u) >> m->table.shift, probe;
        for (probe = 0; probe <= mask; probe++) {
            if (tab[slot].key == key) {
                uint32_t len = tab[slot].len, e;
                tmp = (uint32_t *)realloc(tmp, sizeof(uint32_t) * (len ? len : 1));
                tmi = (uint16_t *)realloc(tmi, sizeof(uint16_t) * (len ? len : 1));
                HIPCHK(hipMemcpy(tmp, m->ent.e4 + tab[slot].start, sizeof(uint32_t) * len, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(tmi, m->ent.mi + tab[slot].start, sizeof(uint16_t) * len, hipMemcpyDeviceToHost));
                for (e = 0; e < len; e++, total++)
                    if (pairs_out && written < cap)
                        pairs_out[written++] = ((uint32_t)s * OSLAMK_SLICE + pc_local_of_row11(tmp[e] & PC_ROW_MASK)) * (uint32_t)m->c.n + tmi[e];
                break;
            }
            if (tab[slot].key == 0) break;
            slot = (slot + 1) & mask;
        }
    }
    if (pairs_out) qsort(pairs_out, written, sizeof(uint32_t), u32_order);
    *count_out = total;
done:
    free(tmp);
    free(tmi);
    return rc;
}

int oslam_model_bucket_words(oslam_model *m, uint32_t key, int slice, uint32_t *words_out, size_t cap, size_t *count_out)
{
    int rc = OSLAM_OK;
    oslamk_slot *tab = NULL;
    uint32_t mask, slot, probe;
    if (!m || !count_out || slice < 0 || slice >= m->table.n_slices) return fail(OSLAM_E_INVALID, "bad argument");
    *count_out = 0;
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    tab = (oslamk_slot *)malloc(sizeof(oslamk_slot) * m->table.cap);
    if (!tab) return fail(OSLAM_E_NOMEM, "host allocation failed");
    HIPCHK(hipMemcpy(tab, m->table.slots + (size_t)slice * m->table.cap, sizeof(oslamk_slot) * m->table.cap, hipMemcpyDeviceToHost));
    mask = m->table.cap - 1;
    slot = (key * 2654435761u) >> m->table.shift;
    for (probe = 0; probe <= mask; probe++, slot = (slot + 1) & mask) {
        if (tab[slot].key == key) {
            size_t n = tab[slot].len < cap ? tab[slot].len : cap;
            *count_out = tab[slot].len;
            if (words_out && n) HIPCHK(hipMemcpy(words_out, m->ent.e4 + tab[slot].start, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
            break;
        }
        if (tab[slot].key == 0) break;
    }
done:
    free(tab);
    return rc;
}

int oslam_vote_accumulator(oslam_model *m, oslam_scene *s, size_t ref_index, uint32_t *acc_out)
{
    int rc = OSLAM_OK;
    uint32_t *d_dump = NULL, *d_ref = NULL, *h_dump = NULL;
    float *d_tsg = NULL;
    float rows[8];
    uint32_t ref = (uint32_t)ref_index;
    oslamk_counters cnt;
    scratch_pool *pool = NULL;
    size_t cells;
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (!acc_out || ref_index >= (size_t)s->c.n) return fail(OSLAM_E_INVALID, "bad reference index");
    if (hipSetDevice(m->dev) != hipSuccess) return fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    cells = (size_t)m->table.n_slices * OSLAMK_SLICE * OSLAMK_NBIN;
    oslam_T_g_rows(s->c.h_xyz, s->c.h_nrm, &ref, 1, rows);
    pool = pool_lock(m->dev);
    if (!pool) return fail(OSLAM_E_LIMIT, "device ordinal too large");
    HIPCHK(hipMalloc((void **)&d_dump, sizeof(uint32_t) * cells));
    HIPCHK(hipMalloc((void **)&d_ref, sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&d_tsg, sizeof rows));
    HIPCHK(hipMemcpy(d_ref, &ref, sizeof ref, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tsg, rows, sizeof rows, hipMemcpyHostToDevice));
    /* one reference point through the same kernels; fixed_gmax = all ones: nothing is emitted */
    rc = run_votes_refs(pool, m, s, d_ref, d_tsg, 1, 0xffffffffu, d_dump, &cnt, NULL, NULL, NULL, NULL, NULL);
    if (rc != OSLAM_OK) goto done;
    h_dump = (uint32_t *)malloc(sizeof(uint32_t) * cells);
    if (!h_dump) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    HIPCHK(hipMemcpy(h_dump, d_dump, sizeof(uint32_t) * cells, hipMemcpyDeviceToHost));
    memcpy(acc_out, h_dump, sizeof(uint32_t) * OSLAMK_NBIN * (size_t)m->c.n);
done:
    pool_unlock(pool);
    free(h_dump);
    if (d_dump) (void)hipFree(d_dump);
    if (d_ref) (void)hipFree(d_ref);
    if (d_tsg) (void)hipFree(d_tsg);
    return rc;
}

int oslam_last_result(oslam_model *m, oslam_scene *s, float *trans_out, float *rots_out, float *vote_counts_out,
                      size_t cap, size_t *n_out, uint32_t *max_idx_out)
{
    int rc;
    size_t n;
    float T[16], *tr = NULL, *ro = NULL, *sc = NULL;
    uint32_t best = 0;
    if (!n_out) return fail(OSLAM_E_INVALID, "NULL argument");
    *n_out = 0;
    rc = check_pair(m, s);
    if (rc != OSLAM_OK) return rc;
    if (materialise_last(m) != OSLAM_OK) return OSLAM_E_DEVICE;
    n = m->n_last;
    if (max_idx_out) *max_idx_out = 0;
    if (n == 0) return OSLAM_OK;
    tr = (float *)calloc(3 * n, sizeof(float));
    ro = (float *)calloc(4 * n, sizeof(float));
    sc = (float *)calloc(n, sizeof(float));
    if (!tr || !ro || !sc) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    rc = oslam_pose_stage_ex(m->last_cells, n, m->c.h_xyz, m->c.h_nrm, (size_t)m->c.n, s->c.h_xyz, s->c.h_nrm, (size_t)s->c.n,
                             m->d_dist, m->params.cpu_clustering, m->params.use_l1_norm, m->params.use_averaged_clusters,
                             m->weights, T, NULL, tr, ro, sc, &best);
    if (rc != OSLAM_OK) { rc = fail(rc, "pose stage failed"); goto done; }
    if (n > cap) n = cap;
    if (trans_out) memcpy(trans_out, tr, sizeof(float) * 3 * n);
    if (rots_out) memcpy(rots_out, ro, sizeof(float) * 4 * n);
    if (vote_counts_out) memcpy(vote_counts_out, sc, sizeof(float) * n);
    if (max_idx_out) *max_idx_out = best;
    *n_out = n;
done:
    free(tr);
    free(ro);
    free(sc);
    return rc;
}

int oslam_last_cells(oslam_model *m, oslam_cell *cells_out, float *poses_out, size_t cap, size_t *n_out)
{
    size_t n;
    if (!m || !n_out) return fail(OSLAM_E_INVALID, "NULL argument");
    if ((cells_out || poses_out) && materialise_last(m) != OSLAM_OK) return OSLAM_E_DEVICE;
    n = m->n_last < cap ? m->n_last : cap;
    if (cells_out) memcpy(cells_out, m->last_cells, sizeof(oslam_cell) * n);
    if (poses_out) memcpy(poses_out, m->last_poses, sizeof(float) * 16 * n);
    *n_out = m->n_last;
    return OSLAM_OK;
}

/* ------------------------------------------------------------------------ */
static uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int oslam_selftest_math(size_t n, uint64_t seed, uint64_t *mismatches)
{
    int rc = OSLAM_OK, dev;
    float *h = NULL, *d = NULL, *ho = NULL;
    size_t i;
    uint64_t bad = 0;
    if (!mismatches || n == 0) return fail(OSLAM_E_INVALID, "bad arguments");
    rc = pick_device(0, &dev);
    if (rc != OSLAM_OK) return rc;
    h = (float *)malloc(sizeof(float) * 3 * n);
    ho = (float *)malloc(sizeof(float) * 4 * n);
    if (!h || !ho) { rc = fail(OSLAM_E_NOMEM, "host allocation failed"); goto done; }
    for (i = 0; i < n; i++) {
        uint64_t r = sm64(&seed), r2 = sm64(&seed);
        float x, y, x2;
        switch (i & 3) {
        case 0:   /* acos domain, dense near +-1 and +-0.5 */
            x = (float)((double)(int32_t)(uint32_t)r / 2147483648.0);
            y = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0 * 3.0);
            x2 = (float)((double)(int32_t)(uint32_t)r2 / 2147483648.0 * 3.0);
            break;
        case 1:
            x = 1.0f - (float)((double)(uint32_t)r / 4294967296.0) * 1e-3f;
            if (r2 & 1) x = -x;
            y = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0);
            x2 = y * ((r2 & 2) ? 0.4375f : 2.4375f) * (1.0f + (float)(int)((r2 >> 8) & 15) * 1e-7f);
            break;
        case 2:   /* raw bit patterns */
            x = PM_BITS_U2F((uint32_t)r);
            y = PM_BITS_U2F((uint32_t)(r >> 32));
            x2 = PM_BITS_U2F((uint32_t)r2);
            break;
        default:
            x = (float)((double)(int32_t)(uint32_t)r / 2147483648.0 * 1.00001);
            y = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0 * 1e-3);
            x2 = (float)((double)(int32_t)(uint32_t)r2 / 2147483648.0 * 1e3);
            break;
        }
        h[i] = x; h[n + i] = y; h[2 * n + i] = x2;
    }
    HIPCHK(hipMalloc((void **)&d, sizeof(float) * 7 * n));
    HIPCHK(hipMemcpy(d, h, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    KCHK(oslamk_selftest(d, d + n, d + 2 * n, n, d + 3 * n, d + 4 * n, (uint32_t *)(d + 5 * n),
                         (uint32_t *)(d + 6 * n), g_stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)g_stream));
    HIPCHK(hipMemcpy(ho, d + 3 * n, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
    for (i = 0; i < n; i++) {
        float x = h[i], y = h[n + i], x2 = h[2 * n + i];
        float a = pm_acosf(x), t = pm_atan2f(y, x2);
        float st = 0.0371f + pm_fabsf(x) * 0.01f;
        uint32_t q = pc_quant_bits(pm_fabsf(y) * 7.0f, st, 1.0f / st);
        uint32_t b = pc_alpha_bin_exact(y, x2, x, y - x2);
        uint32_t ga = PM_BITS_F2U(ho[i]), gt = PM_BITS_F2U(ho[n + i]);
        int a_ok = pm_isnan(a) ? pm_isnan(ho[i]) : (PM_BITS_F2U(a) == ga);
        int t_ok = pm_isnan(t) ? pm_isnan(ho[n + i]) : (PM_BITS_F2U(t) == gt);
        if (!a_ok || !t_ok || q != ((uint32_t *)ho)[2 * n + i] || b != ((uint32_t *)ho)[3 * n + i]) bad++;
    }
    *mismatches = bad;
done:
    free(h);
    free(ho);
    if (d) (void)hipFree(d);
    return rc;
}
