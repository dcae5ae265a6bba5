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

/* The order of the kept cells is (count descending, code ascending).  When the bits fit, that is ONE ascending sort of
 * keys (max count - count) << cb | scene reference << lb | model reference << 6 | alpha -- the code without the unused
 * bits between its halves, so that the radix sort has as few digits as the clouds' sizes allow -- and count and code
 * come back out of the sorted key; otherwise two stable sorts of (code, count) pairs (cb == 0). */
struct pose_pack {
    uint32_t lb, cb, gmax;             /* cb == 0: not packed */
};
__device__ __forceinline__ void pose_unpack(const pose_pack pk, unsigned long long key, unsigned long long *code, uint32_t *count)
{
    const unsigned long long c = key & ((1ull << pk.cb) - 1ull);
    *code = ((c >> pk.lb) << 32) | (c & ((1ull << pk.lb) - 1ull));
    *count = pk.gmax - (uint32_t)(key >> pk.cb);
}

__global__ void k_pose_split(const oslamk_cell *cells, uint32_t n, unsigned long long *code, uint32_t *count,
                             unsigned long long *whole, pose_pack pk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) whole[0] = whole[1] = whole[2] = 0ull;   /* k_pose_cells adds to [0], [1], two sorts later; [2]: k_pose_best's ticket */
    if (i >= n) return;
    const unsigned long long cd = cells[i].code;
    const uint32_t cnt = cells[i].count;
    if (pk.cb) {
        const uint32_t down = cnt < pk.gmax ? pk.gmax - cnt : 0u;
        code[i] = ((unsigned long long)down << pk.cb) | ((cd >> 32) << pk.lb) | (cd & 0xffffffffull);
    } else {
        code[i] = cd;
        count[i] = cnt;
    }
}

/* K5 + K6 + K7 + K8 for cell i of the ordered list */
__global__ __launch_bounds__(256) void k_pose_cells(const uint32_t *count, const unsigned long long *code, uint32_t n,
                                                    const float *Tm16, const float *Ts16, uint32_t df,
                                                    const float *weights, const float *rotx_cs, float d_dist,
                                                    oslamk_cell *cells_out, float *poses, float *trans, float *quat,
                                                    int *cell, float *wv, uint32_t *hash, uint32_t *idx,
                                                    unsigned long long *whole, pose_pack pk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cd = 0;
    uint32_t cnt = 0;
    if (i < n) {
        if (pk.cb) pose_unpack(pk, code[i], &cd, &cnt);
        else { cd = code[i]; cnt = count[i]; }
    }
    {
        /* are the weighted votes whole numbers, and what is their sum (oslamk_cluster_scores: the order-free sum)?
         * one atomic per wave */
        unsigned long long v = 0;
        bool frac = false;
        if (i < n) {
            const float w = weights[((uint32_t)cd) >> 6] * cnt;
            if (w >= 0.0f && w < 16777216.0f && w == (float)(uint32_t)w) v = (uint32_t)w;
            else frac = true;
        }
        __shared__ unsigned long long s_part[4];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            v = s_part[0] + s_part[1] + s_part[2] + s_part[3];
            if (v) atomicAdd(&whole[0], v);
        }
        if (frac) atomicOr(&whole[1], 1ull);
    }
    if (i >= n) return;
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

/* the winner (model.cu:292-295: the first pose with the highest score) and its matrix with the translation of the
 * clustering stage (ppf.cu:86-91), in one record for one copy to the host: out[0] = index, out[1..16] = T */
#define BEST_BLOCKS 64
__global__ __launch_bounds__(1024) void k_pose_best(const float *score, uint32_t n, const float *poses, const float *trans,
                                                    float *out, float *part_v, uint32_t *part_i, unsigned long long *ticket)
{
    __shared__ float s_v[16];
    __shared__ uint32_t s_i[16];
    __shared__ bool s_last;
    float bv = -1.0f;                                /* scores are >= 1 */
    uint32_t bi = 0xffffffffu;
    auto better = [](float ov, uint32_t oi, float v, uint32_t i) { return ov > v || (ov == v && oi < i); };
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += 1024u * gridDim.x) {
        const float v = score[i];
        if (v > bv) { bv = v; bi = i; }              /* ascending i per thread: the first of equals stays */
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63u) == 0u) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++)
            if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
        part_v[blockIdx.x] = bv;
        part_i[blockIdx.x] = bi;
        __threadfence();
        s_last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull;
        if (s_last) {                                /* the block that finishes last picks among the blocks' winners */
            __threadfence();
            bv = -1.0f;
            bi = 0xffffffffu;
            for (uint32_t b = 0; b < gridDim.x; b++) {
                const float v = ((volatile float *)part_v)[b];
                const uint32_t i = ((volatile uint32_t *)part_i)[b];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            reinterpret_cast<uint32_t *>(out)[0] = bi;
            for (int k = 0; k < 16; k++) out[1 + k] = poses[16 * (size_t)bi + k];
            out[1 + 3] = trans[3 * (size_t)bi];
            out[1 + 7] = trans[3 * (size_t)bi + 1];
            out[1 + 11] = trans[3 * (size_t)bi + 2];
        }
    }
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
 * record on every call costs more than the kernels at 10^5 records).  The tails of several models can be in
 * flight on the stream at once (a database frame: all selections, one wait, all chains, one wait): the chains share
 * the work space -- they run one after the other, in stream order -- and every one has a slot for what goes back to
 * the host: the number of selected records, the winner's index and matrix. */
#define POSE_SLOT_WORDS 32
struct pose_pool {
    char *d;
    size_t cap;
    void *tmp;
    size_t tmp_cap, tmp_sel, tmp_s64, tmp_s32, tmp_k64;
    uint32_t slots;
    float *h_pin;              /* pinned [slots][POSE_SLOT_WORDS]: word 0 the count, words 1.. the winner's record */
    float *d_small;            /* [0..127] the rotation table, then [slots][POSE_SLOT_WORDS] */
    int rot_loaded;
};
static pose_pool g_pool[64];

/* the arrays of one chain over N records, carved out of the pool */
struct pose_carve {
    size_t codeA, codeB, cntA, cntB, trans, quat, cell, wv, hash, idx, shash, sidx, sq, st, sw, score, whole, part, tab, total;
};
static void carve_for(size_t N, pose_carve *c)
{
    size_t off = 0;
    c->codeA = off; off += align_up(8 * N);
    c->codeB = off; off += align_up(8 * N);
    c->cntA = off; off += align_up(4 * N);
    c->cntB = off; off += align_up(4 * N);
    c->trans = off; off += align_up(12 * N);
    c->quat = off; off += align_up(16 * N);
    c->cell = off; off += align_up(12 * N);
    c->wv = off; off += align_up(4 * N);
    c->hash = off; off += align_up(4 * N);
    c->idx = off; off += align_up(4 * N);
    c->shash = off; off += align_up(4 * N);
    c->sidx = off; off += align_up(4 * N);
    c->sq = off; off += align_up(16 * N);
    c->st = off; off += align_up(12 * N);
    c->sw = off; off += align_up(4 * N);
    c->score = off; off += align_up(4 * N);
    c->whole = off; off += align_up(32);
    c->part = off; off += align_up(8 * BEST_BLOCKS);
    c->tab = off; off += align_up(4 * oslamk_cluster_table_words((int)N));
    c->total = off;
}

static pose_pool *cur_pool(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NULL;
    return &g_pool[dev];
}

/* work space for chains over up to n_max records and `slots` chains in flight; waits for the device when it has to
 * grow (nothing of an earlier call may still be running: callers reserve before they enqueue) */
extern "C" int oslamk_pose_reserve(uint32_t n_max, uint32_t slots, const float *h_rotx_cs, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    pose_pool *p = cur_pool();
    int rc = 0;
    pose_carve c;
    size_t tmp_sel = 0, tmp_s64 = 0, tmp_s32 = 0, tmp_k64 = 0, tmp_bytes;
    const size_t N = n_max ? n_max : 1;
    if (!p) return (int)hipErrorInvalidDevice;
    if (slots == 0) slots = 1;
    carve_for(N, &c);
    PCHK(rocprim::select(nullptr, tmp_sel, (const oslamk_cell *)nullptr, (oslamk_cell *)nullptr, (uint32_t *)nullptr, N, cell_above{0.0f}, stream));
    PCHK(rocprim::radix_sort_pairs(nullptr, tmp_s64, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                   (uint32_t *)nullptr, (uint32_t *)nullptr, N, 0, 64, stream));
    PCHK(rocprim::radix_sort_pairs_desc(nullptr, tmp_s32, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                        (unsigned long long *)nullptr, (unsigned long long *)nullptr, N, 0, 32, stream));
    PCHK(rocprim::radix_sort_keys(nullptr, tmp_k64, (unsigned long long *)nullptr, (unsigned long long *)nullptr, N, 0, 64, stream));
    tmp_bytes = tmp_sel > tmp_s64 ? tmp_sel : tmp_s64;
    tmp_bytes = tmp_bytes > tmp_s32 ? tmp_bytes : tmp_s32;
    tmp_bytes = tmp_bytes > tmp_k64 ? tmp_bytes : tmp_k64;
    if (tmp_bytes < 16) tmp_bytes = 16;
    if (p->slots < slots) {
        const uint32_t want = slots < 64 ? 64 : slots + slots / 2;
        if (p->h_pin) (void)hipHostFree(p->h_pin);
        if (p->d_small) (void)hipFree(p->d_small);
        p->h_pin = NULL;
        p->d_small = NULL;
        p->slots = 0;
        p->rot_loaded = 0;
        PCHK(hipHostMalloc((void **)&p->h_pin, sizeof(float) * POSE_SLOT_WORDS * want, hipHostMallocDefault));
        PCHK(hipMalloc((void **)&p->d_small, sizeof(float) * (128 + (size_t)POSE_SLOT_WORDS * want)));
        p->slots = want;
    }
    if (p->cap < c.total) {
        if (p->d) (void)hipFree(p->d);
        p->d = NULL;
        p->cap = 0;
        PCHK(hipMalloc((void **)&p->d, c.total + c.total / 4));
        p->cap = c.total + c.total / 4;
    }
    if (p->tmp_cap < tmp_bytes) {
        if (p->tmp) (void)hipFree(p->tmp);
        p->tmp = NULL;
        p->tmp_cap = 0;
        PCHK(hipMalloc(&p->tmp, tmp_bytes + tmp_bytes / 4));
        p->tmp_cap = tmp_bytes + tmp_bytes / 4;
    }
    /* rocprim is handed the size its query returned for n_max; a smaller problem needs no more */
    p->tmp_sel = tmp_sel;
    p->tmp_s64 = tmp_s64;
    p->tmp_s32 = tmp_s32;
    p->tmp_k64 = tmp_k64;
    if (!p->rot_loaded && h_rotx_cs) {                 /* the table is the same for every call (oslam_rotx_table) */
        PCHK(hipMemcpyAsync(p->d_small, h_rotx_cs, 128 * sizeof(float), hipMemcpyHostToDevice, stream));
        PCHK(hipStreamSynchronize(stream));            /* the source is pageable host memory */
        p->rot_loaded = 1;
    }
done:
    return rc;
}

/* the records with count > min_votecount, compacted in their order into d_sel (device, capacity n_in); their number
 * goes to the slot (oslamk_pose_selected, after the stream has been waited for) */
extern "C" int oslamk_pose_select_async(const oslamk_cell *d_in, uint32_t n_in, float min_votecount, oslamk_cell *d_sel,
                                        uint32_t slot, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    pose_pool *p = cur_pool();
    int rc = 0;
    size_t tmp;
    if (!p || slot >= p->slots) return (int)hipErrorInvalidValue;
    uint32_t *d_count = (uint32_t *)(p->d_small + 128 + (size_t)POSE_SLOT_WORDS * slot);
    tmp = p->tmp_sel;
    if (n_in == 0) {
        PCHK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
    } else {
        PCHK(rocprim::select(p->tmp, tmp, d_in, d_sel, d_count, (size_t)n_in, cell_above{min_votecount}, stream));
    }
    PCHK(hipMemcpyAsync(p->h_pin + (size_t)POSE_SLOT_WORDS * slot, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
done:
    return rc;
}

extern "C" uint32_t oslamk_pose_selected(uint32_t slot)
{
    pose_pool *p = cur_pool();
    uint32_t n = 0;
    if (p && slot < p->slots) memcpy(&n, p->h_pin + (size_t)POSE_SLOT_WORDS * slot, sizeof n);
    return n;
}

/* The chain over the n >= 2 selected records in d_sel: order, poses, clustering scores, winner.  d_cells_out (may be
 * d_sel itself) = the cells in (count desc, code asc) order, d_poses = 16 floats per cell; the winner's index and its
 * matrix with the translation of the clustering stage (ppf.cu:74-93) go to the slot (oslamk_pose_result). */
static uint32_t bits_for(uint64_t max_value)
{
    uint32_t b = 0;
    while (max_value >> b) b++;
    return b;
}

extern "C" int oslamk_pose_finish_async(uint32_t n, const oslamk_cell *d_sel, const float *d_Tm16, const float *d_Ts16,
                                        uint32_t df, const float *d_weights, float d_dist, int use_l1,
                                        oslamk_cell *d_cells_out, float *d_poses, uint32_t gmax, uint32_t model_points,
                                        uint32_t scene_points, int two_sorts, uint32_t slot, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    pose_pool *p = cur_pool();
    int rc = 0;
    pose_carve c;
    if (!p || slot >= p->slots || n < 2) return (int)hipErrorInvalidValue;
    carve_for(n, &c);
    if (c.total > p->cap) return (int)hipErrorInvalidValue;        /* oslamk_pose_reserve was given less */
    {
        char *d = p->d;
        void *d_tmp = p->tmp;
        size_t tmp_s64 = p->tmp_s64, tmp_s32 = p->tmp_s32;
        unsigned long long *codeA = (unsigned long long *)(d + c.codeA), *codeB = (unsigned long long *)(d + c.codeB);
        uint32_t *cntA = (uint32_t *)(d + c.cntA), *cntB = (uint32_t *)(d + c.cntB);
        float *trans = (float *)(d + c.trans), *quat = (float *)(d + c.quat), *wv = (float *)(d + c.wv);
        int *cell = (int *)(d + c.cell);
        uint32_t *hash = (uint32_t *)(d + c.hash), *idx = (uint32_t *)(d + c.idx);
        uint32_t *shash = (uint32_t *)(d + c.shash), *sidx = (uint32_t *)(d + c.sidx);
        float *sq = (float *)(d + c.sq), *st = (float *)(d + c.st), *sw = (float *)(d + c.sw), *score = (float *)(d + c.score);
        unsigned long long *whole = (unsigned long long *)(d + c.whole);
        float *rot = p->d_small, *d_best = p->d_small + 128 + (size_t)POSE_SLOT_WORDS * slot + 1;
        const unsigned blocks = (n + 255) / 256;
        const unsigned bb = (n + 4095u) / 4096u < BEST_BLOCKS ? (n + 4095u) / 4096u : BEST_BLOCKS;
        /* (count desc, code asc): one sort of packed keys when their bits fit (see pose_pack), else a stable sort by
         * code and then a stable descending sort by count */
        pose_pack pk = {0, 0, gmax};
        {
            const uint32_t lb = 6 + bits_for(model_points ? model_points - 1u : 0u), sb = bits_for(scene_points ? scene_points - 1u : 0u);
            if (!two_sorts && lb <= 32 && lb + sb + bits_for(gmax) <= 64) { pk.lb = lb; pk.cb = lb + sb; }
        }
        hipLaunchKernelGGL(k_pose_split, dim3(blocks), dim3(256), 0, stream, d_sel, n, codeA, cntA, whole, pk);
        if (pk.cb) {
            size_t tmp_k = p->tmp_k64;
            PCHK(rocprim::radix_sort_keys(d_tmp, tmp_k, codeA, codeB, (size_t)n, 0, pk.cb + bits_for(gmax), stream));
            unsigned long long *sw_ = codeA; codeA = codeB; codeB = sw_;      /* the sorted keys are read from codeA below */
        } else {
            PCHK(rocprim::radix_sort_pairs(d_tmp, tmp_s64, codeA, codeB, cntA, cntB, (size_t)n, 0, 64, stream));
            PCHK(rocprim::radix_sort_pairs_desc(d_tmp, tmp_s32, cntB, cntA, codeB, codeA, (size_t)n, 0, 32, stream));
        }
        hipLaunchKernelGGL(k_pose_cells, dim3(blocks), dim3(256), 0, stream, cntA, codeA, n, d_Tm16, d_Ts16, df, d_weights,
                           rot, d_dist, d_cells_out, d_poses, trans, quat, cell, wv, hash, idx, whole, pk);
        /* (cell key, index) ascending: indices are ascending already and the sort is stable */
        PCHK(rocprim::radix_sort_pairs(d_tmp, tmp_s32, hash, shash, idx, sidx, (size_t)n, 0, 32, stream));
        hipLaunchKernelGGL(k_pose_gather, dim3(blocks), dim3(256), 0, stream, sidx, n, quat, trans, wv, sq, st, sw);
        rc = oslamk_cluster_scores((int)n, cell, shash, sidx, sq, st, sw, d_dist, use_l1, score, 0, whole,
                                   (uint32_t *)(d + c.tab), stream_);
        if (rc != 0) goto done;
        hipLaunchKernelGGL(k_pose_best, dim3(bb), dim3(1024), 0, stream, score, n, d_poses, trans, d_best,
                           (float *)(d + c.part), (uint32_t *)(d + c.part + 4 * BEST_BLOCKS), whole + 2);
        PCHK(hipMemcpyAsync(p->h_pin + (size_t)POSE_SLOT_WORDS * slot + 1, d_best, 17 * sizeof(float), hipMemcpyDeviceToHost, stream));
        PCHK(hipGetLastError());
    }
done:
    return rc;
}

extern "C" void oslamk_pose_result(uint32_t slot, uint32_t *best_out, float T_best[16])
{
    pose_pool *p = cur_pool();
    if (!p || slot >= p->slots) return;
    memcpy(best_out, p->h_pin + (size_t)POSE_SLOT_WORDS * slot + 1, sizeof(uint32_t));
    memcpy(T_best, p->h_pin + (size_t)POSE_SLOT_WORDS * slot + 2, 16 * sizeof(float));
}

/* the records with count > min_votecount, compacted in their order into d_out (device, capacity n_in) */
extern "C" int oslamk_select_cells(const oslamk_cell *d_in, uint32_t n_in, float min_votecount, oslamk_cell *d_out,
                                   uint32_t *n_out, void *stream_)
{
    int rc = 0;
    *n_out = 0;
    if (n_in == 0) return 0;
    rc = oslamk_pose_reserve(n_in, 1, NULL, stream_);
    if (rc != 0) return rc;
    rc = oslamk_pose_select_async(d_in, n_in, min_votecount, d_out, 0, stream_);
    if (rc != 0) return rc;
    PCHK(hipStreamSynchronize((hipStream_t)stream_));
    *n_out = oslamk_pose_selected(0);
done:
    return rc;
}

/* frees the tail's work space on the current device */
extern "C" void oslamk_pose_release(void)
{
    pose_pool *p = cur_pool();
    if (!p) return;
    if (p->d) (void)hipFree(p->d);
    if (p->tmp) (void)hipFree(p->tmp);
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    if (p->d_small) (void)hipFree(p->d_small);
    memset(p, 0, sizeof *p);
}

/* One model's tail, start to end.  d_cells_in[n_in]: emitted peak records (device).  Outputs (device, caller-owned,
 * capacity n_in): d_cells_out = the kept cells in (count desc, code asc) order, d_poses = 16 floats per kept cell.
 * Host outputs: *n_out kept cells, *best_out index of the winning pose, T_best its matrix with the
 * translation of the clustering stage (ppf.cu:74-93).  Returns a hipError_t as int. */
extern "C" int oslamk_pose_stage(const oslamk_cell *d_cells_in, uint32_t n_in, float min_votecount, const float *d_Tm16,
                                 const float *d_Ts16, uint32_t df, const float *d_weights, const float *h_rotx_cs,
                                 float d_dist, int use_l1, oslamk_cell *d_cells_out, float *d_poses, uint32_t gmax,
                                 uint32_t model_points, uint32_t scene_points, int two_sorts, uint32_t *n_out,
                                 uint32_t *best_out, float T_best[16], void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = 0;
    uint32_t n;
    *n_out = 0;
    *best_out = 0;
    if (n_in == 0) return 0;
    rc = oslamk_pose_reserve(n_in, 1, h_rotx_cs, stream_);
    if (rc != 0) return rc;
    rc = oslamk_pose_select_async(d_cells_in, n_in, min_votecount, d_cells_out, 0, stream_);
    if (rc != 0) return rc;
    PCHK(hipStreamSynchronize(stream));
    n = oslamk_pose_selected(0);
    *n_out = n;
    if (n < 2) return 0;                               /* the caller's host path handles 0 and 1 cells */
    rc = oslamk_pose_finish_async(n, d_cells_out, d_Tm16, d_Ts16, df, d_weights, d_dist, use_l1, d_cells_out, d_poses, gmax,
                                  model_points, scene_points, two_sorts, 0, stream_);
    if (rc != 0) return rc;
    PCHK(hipStreamSynchronize(stream));
    oslamk_pose_result(0, best_out, T_best);
done:
    return rc;
}
