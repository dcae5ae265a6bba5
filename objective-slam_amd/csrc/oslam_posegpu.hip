/*
 * oslam_posegpu.hip -- the pose tail on the GPU for large peak sets: what the reference runs as
 * K5..K9 (pcl/alignment/src/cuda/kernel.cu:605-782, model.cu:148-244,292-295) after the votes.
 * The host version (oslam_pose.c) is exact and fine for a few thousand peaks; a model that is
 * absent from the scene, a planar scene or the union of eight GPUs' peaks can leave 10^5..10^6
 * cells above 0.4 * max, and then the host sort, poses and clustering take longer than the votes.
 *
 *   filter    count > min_votecount (model.cu:164-167)                rocPRIM select
 *   order     count descending, code ascending (the host's order)     two stable radix sorts
 *   K5..K8    pose, weighted votes, quaternion, translation cell + key one thread per cell
 *   K9        (cell key, index) radix sort, gather, k_cluster_scores   (oslam_kernels.hip)
 *   argmax    first maximum of the scores (model.cu:292-295)           on the host over n floats
 *
 * Every float sequence is the host's (oslam_pose_math.h, -ffp-contract=off); the trigonometry is
 * not redone here: the frames T_g (libm on the host, one per point) and the 64 rotations about x
 * come in as tables.  Results are therefore the host's bit for bit (GPU tests force this path at
 * small sizes and compare with the oracle).
 */
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <stdint.h>
#include <stdlib.h>

#include "oslam_kernels.h"
#include "oslam_pose_math.h"

struct cell_above {
    float min_votecount;
    __device__ bool operator()(const oslamk_cell &c) const { return (float)c.count > min_votecount; }
};

__global__ void k_pose_split(const oslamk_cell *cells, uint32_t n, unsigned long long *code, uint32_t *count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    code[i] = cells[i].code;
    count[i] = cells[i].count;
}

/* K5 + K6 + K7 + K8 for cell i of the ordered list */
__global__ __launch_bounds__(256) void k_pose_cells(const uint32_t *count, const unsigned long long *code, uint32_t n,
                                                    const float *Tm16, const float *Ts16, uint32_t df,
                                                    const float *weights, const float *rotx_cs, float d_dist,
                                                    oslamk_cell *cells_out, float *poses, float *trans, float *quat,
                                                    int *cell, float *wv, uint32_t *hash, uint32_t *idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long cd = code[i];
    const uint32_t cnt = count[i];
    const uint32_t s_r = (uint32_t)(cd >> 32), lo = (uint32_t)cd, m_r = lo >> 6, a = lo & 63u;
    float T[16];
    if (s_r == 0 && lo == 0) {                      /* kernel.cu:628-631: the (0,0,0) code gets no pose */
        pq_mat_zero(T);
    } else {
        float Tm[16], Ts[16];
        for (int k = 0; k < 16; k++) {
            Tm[k] = Tm16[16 * (size_t)m_r + k];
            Ts[k] = Ts16[16 * (size_t)(s_r / df) + k];
        }
        pq_cell_pose(Tm, Ts, rotx_cs[2 * a], rotx_cs[2 * a + 1], T);
    }
    for (int k = 0; k < 16; k++) poses[16 * (size_t)i + k] = T[k];
    float q[4];
    pq_pose_quat(T, q);
    int32_t c3[3];
    for (int k = 0; k < 3; k++) {
        const float t = T[4 * k + 3];
        trans[3 * (size_t)i + k] = t;
        c3[k] = pq_cell_coord(t, d_dist);
        cell[3 * (size_t)i + k] = c3[k];
    }
    for (int k = 0; k < 4; k++) quat[4 * (size_t)i + k] = q[k];
    wv[i] = weights[m_r] * cnt;                      /* kernel.cu:777 */
    hash[i] = pq_fnv_cell(c3);
    idx[i] = i;
    oslamk_cell oc;
    oc.code = cd;
    oc.count = cnt;
    oc.pad = 0;
    cells_out[i] = oc;
}

/* the pose data permuted into (cell key, index) order for k_cluster_scores */
__global__ void k_pose_gather(const uint32_t *order, uint32_t n, const float *quat, const float *trans, const float *wv,
                              float *sq, float *st, float *sw)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t o = order[j];
    for (int k = 0; k < 4; k++) sq[4 * (size_t)j + k] = quat[4 * (size_t)o + k];
    for (int k = 0; k < 3; k++) st[3 * (size_t)j + k] = trans[3 * (size_t)o + k];
    sw[j] = wv[o];
}

#define PCHK(call)                   \
    do {                             \
        hipError_t e_ = (call);      \
        if (e_ != hipSuccess) {      \
            rc = (int)e_;            \
            goto done;               \
        }                            \
    } while (0)

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

/* work space of the tail, one per device, grown on demand and kept (allocating and freeing ~130 B per
 * record on every call costs more than the kernels at 10^5 records) */
struct pose_pool {
    char *d;
    size_t cap;
    void *tmp;
    size_t tmp_cap;
};
static pose_pool g_pool[64];

static int pool_reserve(pose_pool *p, size_t bytes, size_t tmp_bytes)
{
    if (p->cap < bytes) {
        if (p->d) (void)hipFree(p->d);
        p->d = NULL;
        p->cap = 0;
        hipError_t e = hipMalloc((void **)&p->d, bytes + bytes / 4);
        if (e != hipSuccess) return (int)e;
        p->cap = bytes + bytes / 4;
    }
    if (p->tmp_cap < tmp_bytes) {
        if (p->tmp) (void)hipFree(p->tmp);
        p->tmp = NULL;
        p->tmp_cap = 0;
        hipError_t e = hipMalloc(&p->tmp, tmp_bytes + tmp_bytes / 4);
        if (e != hipSuccess) return (int)e;
        p->tmp_cap = tmp_bytes + tmp_bytes / 4;
    }
    return 0;
}

/* the records with count > min_votecount, compacted in their order into d_out (device, capacity n_in) */
extern "C" int oslamk_select_cells(const oslamk_cell *d_in, uint32_t n_in, float min_votecount, oslamk_cell *d_out,
                                   uint32_t *n_out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = 0, dev = 0;
    size_t tmp = 0;
    uint32_t n = 0;
    pose_pool *pool;
    *n_out = 0;
    if (n_in == 0) return 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    pool = &g_pool[dev];
    PCHK(rocprim::select(nullptr, tmp, d_in, (oslamk_cell *)nullptr, (uint32_t *)nullptr, (size_t)n_in, cell_above{min_votecount}, stream));
    rc = pool_reserve(pool, 256, tmp ? tmp : 16);
    if (rc != 0) goto done;
    PCHK(rocprim::select(pool->tmp, tmp, d_in, d_out, (uint32_t *)pool->d, (size_t)n_in, cell_above{min_votecount}, stream));
    PCHK(hipMemcpyAsync(&n, pool->d, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    PCHK(hipStreamSynchronize(stream));
    *n_out = n;
done:
    return rc;
}

/* frees the tail's work space on the current device */
extern "C" void oslamk_pose_release(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    if (g_pool[dev].d) (void)hipFree(g_pool[dev].d);
    if (g_pool[dev].tmp) (void)hipFree(g_pool[dev].tmp);
    g_pool[dev].d = NULL;
    g_pool[dev].tmp = NULL;
    g_pool[dev].cap = g_pool[dev].tmp_cap = 0;
}

/* d_cells_in[n_in]: emitted peak records (device).  Outputs (device, caller-owned, capacity n_in):
 * d_cells_out = the kept cells in (count desc, code asc) order, d_poses = 16 floats per kept cell.
 * Host outputs: *n_out kept cells, *best_out index of the winning pose, T_best its matrix with the
 * translation of the clustering stage (ppf.cu:74-93).  Returns a hipError_t as int, or -2 when the
 * host could not allocate. */
extern "C" int oslamk_pose_stage(const oslamk_cell *d_cells_in, uint32_t n_in, float min_votecount, const float *d_Tm16,
                                 const float *d_Ts16, uint32_t df, const float *d_weights, const float *h_rotx_cs,
                                 float d_dist, int use_l1, oslamk_cell *d_cells_out, float *d_poses, uint32_t *n_out,
                                 uint32_t *best_out, float T_best[16], void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = 0;
    char *d = NULL;
    void *d_tmp = NULL;
    float *h_score = NULL;
    int dev = 0;
    pose_pool *pool;
    uint32_t n = 0;
    size_t tmp_sel = 0, tmp_s64 = 0, tmp_s32 = 0, tmp_bytes;
    *n_out = 0;
    *best_out = 0;
    if (n_in == 0) return 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    pool = &g_pool[dev];
    {
        /* one allocation, carved */
        const size_t N = n_in;
        size_t off = 0;
        const size_t o_sel = off; off += align_up(sizeof(oslamk_cell) * N);
        const size_t o_cnt = off; off += align_up(4);
        const size_t o_codeA = off; off += align_up(8 * N);
        const size_t o_codeB = off; off += align_up(8 * N);
        const size_t o_cntA = off; off += align_up(4 * N);
        const size_t o_cntB = off; off += align_up(4 * N);
        const size_t o_trans = off; off += align_up(12 * N);
        const size_t o_quat = off; off += align_up(16 * N);
        const size_t o_cell = off; off += align_up(12 * N);
        const size_t o_wv = off; off += align_up(4 * N);
        const size_t o_hash = off; off += align_up(4 * N);
        const size_t o_idx = off; off += align_up(4 * N);
        const size_t o_shash = off; off += align_up(4 * N);
        const size_t o_sidx = off; off += align_up(4 * N);
        const size_t o_sq = off; off += align_up(16 * N);
        const size_t o_st = off; off += align_up(12 * N);
        const size_t o_sw = off; off += align_up(4 * N);
        const size_t o_score = off; off += align_up(4 * N);
        const size_t o_rot = off; off += align_up(128 * 4);
        PCHK(rocprim::select(nullptr, tmp_sel, d_cells_in, (oslamk_cell *)nullptr, (uint32_t *)nullptr, N, cell_above{min_votecount}, stream));
        PCHK(rocprim::radix_sort_pairs(nullptr, tmp_s64, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                       (uint32_t *)nullptr, (uint32_t *)nullptr, N, 0, 64, stream));
        PCHK(rocprim::radix_sort_pairs_desc(nullptr, tmp_s32, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                            (unsigned long long *)nullptr, (unsigned long long *)nullptr, N, 0, 32, stream));
        tmp_bytes = tmp_sel > tmp_s64 ? tmp_sel : tmp_s64;
        tmp_bytes = tmp_bytes > tmp_s32 ? tmp_bytes : tmp_s32;
        rc = pool_reserve(pool, off, tmp_bytes ? tmp_bytes : 16);
        if (rc != 0) goto done;
        d = pool->d;
        d_tmp = pool->tmp;
        oslamk_cell *sel = (oslamk_cell *)(d + o_sel);
        uint32_t *d_count = (uint32_t *)(d + o_cnt);
        unsigned long long *codeA = (unsigned long long *)(d + o_codeA), *codeB = (unsigned long long *)(d + o_codeB);
        uint32_t *cntA = (uint32_t *)(d + o_cntA), *cntB = (uint32_t *)(d + o_cntB);
        float *trans = (float *)(d + o_trans), *quat = (float *)(d + o_quat), *wv = (float *)(d + o_wv);
        int *cell = (int *)(d + o_cell);
        uint32_t *hash = (uint32_t *)(d + o_hash), *idx = (uint32_t *)(d + o_idx);
        uint32_t *shash = (uint32_t *)(d + o_shash), *sidx = (uint32_t *)(d + o_sidx);
        float *sq = (float *)(d + o_sq), *st = (float *)(d + o_st), *sw = (float *)(d + o_sw), *score = (float *)(d + o_score);
        float *rot = (float *)(d + o_rot);
        cell_above pred = {min_votecount};


        PCHK(rocprim::select(d_tmp, tmp_sel, d_cells_in, sel, d_count, N, pred, stream));
        PCHK(hipMemcpyAsync(&n, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        PCHK(hipMemcpyAsync(rot, h_rotx_cs, 128 * sizeof(float), hipMemcpyHostToDevice, stream));
        PCHK(hipStreamSynchronize(stream));
        *n_out = n;
        if (n < 2) goto done;                          /* the caller's host path handles 0 and 1 cells */
        {
            const unsigned blocks = (n + 255) / 256;
            hipLaunchKernelGGL(k_pose_split, dim3(blocks), dim3(256), 0, stream, sel, n, codeA, cntA);
            /* (count desc, code asc): stable sort by code, then stable descending sort by count */
            PCHK(rocprim::radix_sort_pairs(d_tmp, tmp_s64, codeA, codeB, cntA, cntB, (size_t)n, 0, 64, stream));
            PCHK(rocprim::radix_sort_pairs_desc(d_tmp, tmp_s32, cntB, cntA, codeB, codeA, (size_t)n, 0, 32, stream));
            hipLaunchKernelGGL(k_pose_cells, dim3(blocks), dim3(256), 0, stream, cntA, codeA, n, d_Tm16, d_Ts16, df, d_weights,
                               rot, d_dist, d_cells_out, d_poses, trans, quat, cell, wv, hash, idx);
            /* (cell key, index) ascending: indices are ascending already and the sort is stable */
            PCHK(rocprim::radix_sort_pairs(d_tmp, tmp_s32, hash, shash, idx, sidx, (size_t)n, 0, 32, stream));
            hipLaunchKernelGGL(k_pose_gather, dim3(blocks), dim3(256), 0, stream, sidx, n, quat, trans, wv, sq, st, sw);
            rc = oslamk_cluster_scores((int)n, trans, quat, cell, shash, sq, st, sw, d_dist, use_l1, score, stream_);
            if (rc != 0) goto done;
            h_score = (float *)malloc(sizeof(float) * n);
            if (!h_score) { rc = -2; goto done; }
            PCHK(hipMemcpyAsync(h_score, score, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
            PCHK(hipStreamSynchronize(stream));
            {
                uint32_t best = 0, i;
                float tb[3];
                for (i = 1; i < n; i++) if (h_score[i] > h_score[best]) best = i;     /* model.cu:292-295 */
                *best_out = best;
                PCHK(hipMemcpyAsync(T_best, d_poses + 16 * (size_t)best, 16 * sizeof(float), hipMemcpyDeviceToHost, stream));
                PCHK(hipMemcpyAsync(tb, trans + 3 * (size_t)best, 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
                PCHK(hipStreamSynchronize(stream));
                T_best[3] = tb[0]; T_best[7] = tb[1]; T_best[11] = tb[2];             /* ppf.cu:86-91 */
            }
            PCHK(hipGetLastError());
        }
    }
done:
    free(h_score);
    return rc;
}
