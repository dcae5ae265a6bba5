/*
 * oslam_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the PPF registration
 * path.  Wave = 64 lanes throughout.  Built with -ffp-contract=off: the float
 * sequences of ppf_core.h must round exactly as on the host.
 *
 * What the reference does with N^2-sized arrays and Thrust sorts
 * (pcl/alignment/src/cuda/{scene,model}.cu) is fused here:
 *   model build : pair key -> per-slice open-addressing table -> bucketed
 *                 4-byte pair entries (two counting passes, no sort); union table
 *                 of all keys; bitset of the distance bins that can reach a key;
 *   scene keys  : per (reference r, tile of scene points): distance bin of every
 *                 pair, unreachable bins dropped, the rest compacted in LDS; full
 *                 key -> union-table probe -> per-reference hit list
 *                 {key, T_s_g*s_i, theta_v} written by wave-aggregated appends;
 *   voting      : one workgroup per (scene reference point, model slice):
 *                 hit -> slice table probe -> wave-cooperative, prefetched sweep
 *                 of the bucket (16 bytes = 4 entries per lane) -> integer alpha
 *                 bin -> LDS accumulator [1024 model refs][32 alpha bins]; votes
 *                 near a bin edge queued and re-evaluated with the reference's
 *                 float sequence; in-kernel peak extraction.
 */
#include <hip/hip_runtime.h>

#include "oslam_kernels.h"
#include "ppf_core.h"

#define WAVE 64
#ifndef VOTE_PIPE
#define VOTE_PIPE 1       /* chunks in flight ahead of the one being voted (1 measured best) */
#endif
#define VOTE_THREADS 1024
#define ACC_CELLS (OSLAMK_SLICE * OSLAMK_NBIN)

/* thresholds of pc_alpha_bin_table(); every vote workgroup copies them into LDS */
__device__ const uint32_t k_alpha_thr[32] = {PC_ALPHA_THR_FLAT};

__device__ __forceinline__ uint32_t slot_of(uint32_t key, uint32_t shift)
{
    return (key * 2654435761u) >> shift;
}

/* key of the ordered pair (r -> i) of one cloud; r's data is passed in registers */
__device__ __forceinline__ uint32_t cloud_pair_key(const oslamk_cloud &c, int i, float prx,
                                                   float pry, float prz, float nrx, float nry,
                                                   float nrz, float nrn, float d_dist,
                                                   float inv_d_dist, float *pix, float *piy,
                                                   float *piz)
{
    float px = c.px[i], py = c.py[i], pz = c.pz[i];
    float nx = c.nx[i], ny = c.ny[i], nz = c.nz[i];
    *pix = px;
    *piy = py;
    *piz = pz;
    return pc_pair_key(prx, pry, prz, nrx, nry, nrz, nrn, px, py, pz, nx, ny, nz,
                       pc_norm3(nx, ny, nz), d_dist, inv_d_dist);
}

/* --------------------------------------------------------------------------
 * parity tap: keys of one reference row (Scene::getHashKeys row, scene.cu:49-54)
 * ------------------------------------------------------------------------*/
__global__ void k_row_keys(oslamk_cloud c, int ref, float d_dist, float inv_d_dist,
                           uint32_t *keys_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    float nrx = c.nx[ref], nry = c.ny[ref], nrz = c.nz[ref];
    float x, y, z;
    uint32_t k = 0;
    if (i != ref)
        k = cloud_pair_key(c, i, c.px[ref], c.py[ref], c.pz[ref], nrx, nry, nrz,
                           pc_norm3(nrx, nry, nrz), d_dist, inv_d_dist, &x, &y, &z);
    keys_out[i] = k;
}

/* --------------------------------------------------------------------------
 * model build
 * ------------------------------------------------------------------------*/
/* grid (ceil(M/256), M): blockIdx.y = m_r, x covers m_i.  Counts each pair in
 * the table of m_r's slice (Model ctor + ParallelHashArray, model.cu:43-82). */
__global__ void k_model_count(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                              uint32_t *n_unique, uint32_t *overflow)
{
    int m_r = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n || i == m_r) return;
    float nrx = c.nx[m_r], nry = c.ny[m_r], nrz = c.nz[m_r];
    float x, y, z;
    uint32_t key = cloud_pair_key(c, i, c.px[m_r], c.py[m_r], c.pz[m_r], nrx, nry, nrz,
                                  pc_norm3(nrx, nry, nrz), d_dist, inv_d_dist, &x, &y, &z);
    if (key == 0) return;
    int slice = m_r / OSLAMK_SLICE;
    oslamk_slot *tab = t.slots + (size_t)slice * t.cap;
    uint32_t mask = t.cap - 1, slot = slot_of(key, t.shift);
    for (uint32_t probe = 0; probe < t.cap; probe++) {
        uint32_t old = atomicCAS(&tab[slot].key, 0u, key);
        if (old == 0u) atomicAdd(&n_unique[slice], 1u);
        if (old == 0u || old == key) {
            atomicAdd(&tab[slot].len, 1u);
            return;
        }
        slot = (slot + 1) & mask;
    }
    atomicExch(overflow, 1u);
}

/* single workgroup: exclusive scan of len over every slot -> start */
__global__ __launch_bounds__(1024) void k_table_scan(oslamk_table t, uint32_t *total_out)
{
    __shared__ uint32_t part[1024];
    size_t total = (size_t)t.n_slices * t.cap;
    size_t chunk = (total + 1023) / 1024;
    size_t b = (size_t)threadIdx.x * chunk, e = b + chunk < total ? b + chunk : total;
    uint32_t s = 0;
    for (size_t i = b; i < e; i++) s += (t.slots[i].len + 3u) & ~3u;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; i++) {
            uint32_t v = part[i];
            part[i] = run;
            run += v;
        }
        *total_out = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (size_t i = b; i < e; i++) {
        t.slots[i].start = run;
        run += (t.slots[i].len + 3u) & ~3u;
    }
}

/* union table: every key of every slice once (what the scene-key kernel probes);
 * *n_keys counts the distinct keys */
__global__ void k_union_build(oslamk_table t, uint32_t *n_keys, uint32_t *overflow)
{
    size_t total = (size_t)t.n_slices * t.cap;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    uint32_t key = t.slots[idx].key;
    if (key == 0) return;
    uint32_t mask = t.ucap - 1, slot = slot_of(key, t.ushift);
    for (uint32_t probe = 0; probe < t.ucap; probe++) {
        uint32_t old = atomicCAS(&t.ukeys[slot], 0u, key);
        if (old == 0u) atomicAdd(n_keys, 1u);
        if (old == 0u || old == key) return;
        slot = (slot + 1) & mask;
    }
    atomicExch(overflow, 1u);
}

/* one workgroup per distance bin k1: does any of the 17^3 keys of that bin exist in the model? */
__global__ __launch_bounds__(256) void k_reach_build(oslamk_table t, float d_dist)
{
    __shared__ uint32_t s_found;
    const uint32_t k1 = blockIdx.x, mask = t.ucap - 1;
    if (threadIdx.x == 0) s_found = 0;
    __syncthreads();
    for (uint32_t combo = threadIdx.x; combo < PC_ANGLE_COMBOS; combo += 256) {
        const uint32_t key = pc_key_of_bins(k1, combo, d_dist);
        if (key == 0) continue;
        uint32_t slot = slot_of(key, t.ushift);
        for (uint32_t probe = 0; probe <= mask; probe++) {
            const uint32_t k = t.ukeys[slot];
            if (k == key) { s_found = 1; break; }
            if (k == 0) break;
            slot = (slot + 1) & mask;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_found) atomicOr(&t.reach[k1 >> 5], 1u << (k1 & 31u));
}

/* pass 2: same pairs, written into their buckets */
__global__ void k_model_fill(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                             const float *tmg, oslamk_entries ent)
{
    int m_r = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n || i == m_r) return;
    float nrx = c.nx[m_r], nry = c.ny[m_r], nrz = c.nz[m_r];
    float x, y, z;
    uint32_t key = cloud_pair_key(c, i, c.px[m_r], c.py[m_r], c.pz[m_r], nrx, nry, nrz,
                                  pc_norm3(nrx, nry, nrz), d_dist, inv_d_dist, &x, &y, &z);
    if (key == 0) return;
    int slice = m_r / OSLAMK_SLICE;
    oslamk_slot *tab = t.slots + (size_t)slice * t.cap;
    uint32_t mask = t.cap - 1, slot = slot_of(key, t.shift);
    for (uint32_t probe = 0; probe < t.cap; probe++) {
        if (tab[slot].key == key) break;
        slot = (slot + 1) & mask;
    }
    uint32_t pos = atomicAdd(&tab[slot].cur, 1u) & 0x7fffffffu;   /* bit 31 is the marker flag */
    size_t e = (size_t)tab[slot].start + pos;
    const float *rows = tmg + 8 * (size_t)m_r;
    float uy = pc_row_dot(rows, x, y, z), uz = pc_row_dot(rows + 4, x, y, z);
    {
        const uint32_t th = pc_angle_q17(uy, uz);
        /* a marker forces the whole bucket through the exact path: flagged in bit 31 of the cursor */
        if (th == PC_Q17_FORCE) atomicOr(&tab[slot].cur, 0x80000000u);
        ent.e4[e] = ((uint32_t)(m_r - slice * OSLAMK_SLICE) << 22) | (th == PC_Q17_FORCE ? 0u : th);
    }
    ent.mi[e] = (uint16_t)i;
    if (ent.uv) {
        oslamk_uv en;
        en.uy = uy;
        en.uz = uz;
        ent.uv[e] = en;
    }
}

/* --------------------------------------------------------------------------
 * voting
 * ------------------------------------------------------------------------*/
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = __shfl_xor(v, o, WAVE);
        v = v > w ? v : w;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

/* which chunk of which bucket, and which hits share it (all wave-uniform): the hits of a
 * reference point are sorted by key (k_sort_hits), so hits with the same key sit in consecutive
 * lanes and the bucket is streamed once for the whole run */
struct ChunkDesc {
    uint32_t st, off, ln;      /* bucket start, offset of this chunk, bucket length */
    int head, run;             /* first lane of the run and number of hits in it */
    bool valid, bforced;       /* bforced: the bucket holds an entry with the marker */
};

/* A chunk of 256 model-pair entries held in registers by one wave: lane l holds entries
 * 4l .. 4l+3 of the chunk (one 16-byte load).
 * A vote: theta_v - theta_u in units of 2^-17 bin gives bin and position in the bin; only
 * positions within 2^-12 bin of an edge (0.05 % of votes) are re-evaluated with the reference's
 * float sequence via pc_alpha_bin_table (ppf_core.h), so the bins are the reference's.
 * The code is straight-line: lanes past the end of the bucket add into a per-lane trash word
 * behind the accumulator instead of branching around the atomic. */
#define ACC_TRASH ACC_CELLS            /* 64 words, one per lane */

/* Per-wave queue (LDS) of votes to re-evaluate with pc_alpha_bin_table:
 * {entry index, local model reference, v.y, v.z}. */
struct SlowQueue {
    static constexpr uint32_t CAP = 96;
    uint4 *q;
    uint32_t n;                        /* wave-uniform */
    __device__ __forceinline__ void push(unsigned long long mask, bool mine, int lane, uint32_t entry,
                                         uint32_t mr, float vy, float vz)
    {
        if (mine) {
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            q[n + rank] = make_uint4(entry, mr, __builtin_bit_cast(uint32_t, vy), __builtin_bit_cast(uint32_t, vz));
        }
        n += (uint32_t)__popcll(mask);
    }
    __device__ __forceinline__ void flush(const oslamk_vote_args &a, uint32_t *acc, const uint32_t *tbl, int lane)
    {
        for (uint32_t base = 0; base < n; base += WAVE) {
            if (base + lane < n) {
                const uint4 it = q[base + lane];
                const float2 uv = *reinterpret_cast<const float2 *>(&a.ent.uv[it.x]);
                const unsigned bin = pc_alpha_bin_table(uv.x, uv.y, __builtin_bit_cast(float, it.z),
                                                        __builtin_bit_cast(float, it.w), tbl);
                if (bin < OSLAMK_NBIN) atomicAdd(&acc[(it.y << 5) + bin], 1u);
            }
        }
        n = 0;
    }
};

template <int MODE>
struct Chunk {
    uint4 v;
    __device__ __forceinline__ void load(const oslamk_vote_args &a, const ChunkDesc &d, int lane)
    {
        const uint32_t e = d.off + 4u * (uint32_t)lane;
        if (e < d.ln) v = *reinterpret_cast<const uint4 *>(&a.ent.e4[(size_t)d.st + e]);
    }
    /* All hits of the run vote with this chunk.  cs2v / fv / vyv / vzv: per-lane data of the
     * wave's 64 hits (cs2 = (theta_v + 15 bins) mod one turn; fv = theta_v is the marker).
     * Votes that need the reference's float sequence are queued (their operands are a dependent
     * gather that would stall the stream) and evaluated 64 at a time by SlowQueue::flush. */
    __device__ __forceinline__ void vote(const oslamk_vote_args &a, uint32_t *acc, const uint32_t *tbl,
                                         SlowQueue &sq, const ChunkDesc &d, int lane, uint32_t cs2v, bool fv,
                                         float vyv, float vzv) const
    {
        const uint32_t e = d.off + 4u * (uint32_t)lane;
        const int rem = (int)d.ln - (int)e;                  /* entries of this lane that exist */
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t am[4], row[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            am[j] = w[j] & 0x3fffffu;
            /* accumulator row of the entry, or the lane's trash word (then the bin must add 0) */
            row[j] = rem > j ? (w[j] >> 22) << 5 : ACC_TRASH + (uint32_t)lane;
        }
        for (int i = 0; i < d.run; i++) {
            const int li = d.head + i;
            const uint32_t cs2 = readlane_u(cs2v, li);
            const bool forced = d.bforced || (readlane_u((uint32_t)fv, li) != 0);
            uint32_t idx[4];
            bool need[4];
            bool any_need = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t t = cs2 - am[j];                    /* in (-turn, turn) */
                const uint32_t t_wrapped = t + PC_Q17_TURN;
                t = t < t_wrapped ? t : t_wrapped;           /* unsigned min = mod one turn */
                idx[j] = row[j] + (rem > j ? t >> 17 : 0u);
                /* within PC_Q17_MARGIN of a bin edge (either side)?  forced: every vote of the bucket */
                need[j] = MODE == 0 && rem > j &&
                          (forced || ((t - PC_Q17_MARGIN) & (PC_Q17_ONE - 1u)) >= PC_Q17_ONE - 2u * PC_Q17_MARGIN);
                any_need = any_need || need[j];
            }
            if (MODE == 0 && __any(any_need)) {
                const float vy = readlane_f(vyv, li), vz = readlane_f(vzv, li);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned long long nm = __ballot(need[j]);
                    if (nm) {
                        sq.push(nm, need[j], lane, d.st + e + j, w[j] >> 22, vy, vz);
                        if (sq.n > SlowQueue::CAP - WAVE) sq.flush(a, acc, tbl, lane);
                        if (need[j]) idx[j] = ACC_TRASH + (uint32_t)lane;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&acc[idx[j]], 1u);
        }
    }
};

/* Scene::Scene's key pass (scene.cu:24-55: K1 ppf_kernel + K2 ppf_hash_kernel) fused with
 * the lookup of model.cu:96-97, for the pairs (reference point r, point i) of one tile of
 * KEY_TILE scene points.  Phase 1 (cheap, every pair): distance bin only; pairs whose bin
 * cannot produce a model key (table.reach: exact, FNV collisions included) are dropped, the
 * rest are compacted into LDS.  Phase 2 (dense lanes): full key, union-table probe, and pairs
 * that hit are appended to r's hit list as {key, (T_s_g*s_i).y, (T_s_g*s_i).z, theta_v}; one
 * atomic per wave reserves the slots.  grid (ceil(S/KEY_TILE), refs in this batch). */
#define KEY_TILE 4096
__global__ __launch_bounds__(256) void k_scene_hits(oslamk_vote_args a)
{
    __shared__ uint32_t s_list[KEY_TILE];
    __shared__ uint32_t s_n;
    const int ref_local = blockIdx.y;
    const int ref_ord = a.first_ref + ref_local;
    const int lane = threadIdx.x & (WAVE - 1);
    const int S = a.scene.n;
    const uint32_t r = a.ref_idx[ref_ord];
    const float prx = a.scene.px[r], pry = a.scene.py[r], prz = a.scene.pz[r];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();

    for (int c = 0; c < KEY_TILE / 256; c++) {
        const int i = blockIdx.x * KEY_TILE + c * 256 + threadIdx.x;
        bool keep = false;
        if (i < S && (uint32_t)i != r) {
            const int k = pc_pair_dist_bin(a.scene.px[i] - prx, a.scene.py[i] - pry, a.scene.pz[i] - prz,
                                           a.d_dist, a.inv_d_dist);
            keep = (unsigned)k >= OSLAMK_REACH_BINS || ((a.table.reach[k >> 5] >> (k & 31)) & 1u);
        }
        const unsigned long long km = __ballot(keep);
        if (km) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(km));
            base = readlane_u(base, 0);
            if (keep) s_list[base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
    }
    __syncthreads();

    const uint32_t n_keep = s_n;
    const float nrx = a.scene.nx[r], nry = a.scene.ny[r], nrz = a.scene.nz[r];
    const float nrn = pc_norm3(nrx, nry, nrz);
    const float *rows = a.tsg + 8 * (size_t)ref_ord;
    for (uint32_t j0 = 0; j0 < n_keep; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        bool hit = false;
        uint4 rec = make_uint4(0, 0, 0, 0);
        if (j < n_keep) {
            const int i = (int)s_list[j];
            float x, y, z;
            const uint32_t key = cloud_pair_key(a.scene, i, prx, pry, prz, nrx, nry, nrz, nrn, a.d_dist,
                                                a.inv_d_dist, &x, &y, &z);
            if (key != 0) {                                       /* kernel.cu:491,520 */
                const uint32_t mask = a.table.ucap - 1;
                uint32_t slot = slot_of(key, a.table.ushift);
                for (uint32_t probe = 0; probe <= mask; probe++) {
                    const uint32_t k = a.table.ukeys[slot];
                    if (k == key) { hit = true; break; }
                    if (k == 0) break;
                    slot = (slot + 1) & mask;
                }
                if (hit) {
                    const float vy = pc_row_dot(rows, x, y, z);     /* kernel.cu:334-336 */
                    const float vz = pc_row_dot(rows + 4, x, y, z);
                    rec = make_uint4(key, __builtin_bit_cast(uint32_t, vy), __builtin_bit_cast(uint32_t, vz),
                                     pc_angle_q17(vy, vz));
                }
            }
        }
        const unsigned long long hm = __ballot(hit);
        if (hm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.hit_count[ref_local], (uint32_t)__popcll(hm));
            base = readlane_u(base, 0);
            if (hit) {
                const uint32_t rank = (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                reinterpret_cast<uint4 *>(a.hits)[(size_t)ref_local * a.hit_stride + base + rank] = rec;
            }
        }
    }
}

/* Sorts the hit list of each reference point of the batch by key, so that hits that share a
 * bucket are adjacent (on the bench scene a bucket is hit 3.8 times per reference point on
 * average; streaming it once per run cuts the entry traffic 4.7x).  One workgroup per reference
 * point: bitonic sort of (key << 32 | index) in LDS, then the records are gathered into the
 * second list.  Lists longer than SORT_MAX stay in arrival order (still correct, runs are just
 * short). */
#define SORT_MAX 16384
__global__ __launch_bounds__(1024) void k_sort_hits(oslamk_vote_args a)
{
    __shared__ unsigned long long buf[SORT_MAX];
    const int ref_local = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = a.hit_count[ref_local];
    const uint4 *src = reinterpret_cast<const uint4 *>(a.hits) + (size_t)ref_local * a.hit_stride;
    uint4 *dst = reinterpret_cast<uint4 *>(a.hits_sorted) + (size_t)ref_local * a.hit_stride;
    if (n > SORT_MAX) {
        for (uint32_t i = tid; i < n; i += 1024) dst[i] = src[i];
        return;
    }
    uint32_t P = 64;
    while (P < n) P <<= 1;
    for (uint32_t i = tid; i < P; i += 1024)
        buf[i] = i < n ? ((unsigned long long)src[i].x << 32) | i : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < P / 2; t += 1024) {
                /* t-th compare-exchange of this pass: partner indices differ in bit j */
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long x = buf[lo], y = buf[hi];
                const bool up = (lo & k) == 0;
                if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < n; i += 1024) dst[i] = src[(uint32_t)buf[i]];
}

/* One workgroup = one (scene reference point, model slice).
 * LDS: acc[1024][32] u32 = 128 KiB (one workgroup per CU, 16 waves).
 * ComputeUniqueVotes (model.cu:95-171) without the vote list: K3/K4
 * (kernel.cu:480-554) accumulate straight into acc, and the sort/histogram/
 * threshold of model.cu:148-170 becomes the scan at the end. */
template <int MODE>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote(oslamk_vote_args a)
{
    __shared__ uint32_t acc[ACC_CELLS + WAVE];      /* + one trash word per lane */
    __shared__ uint32_t s_wave[VOTE_THREADS / WAVE];
    __shared__ uint32_t s_wave2[VOTE_THREADS / WAVE];
    __shared__ unsigned long long s_wave64[2][VOTE_THREADS / WAVE];
    __shared__ uint32_t s_g, s_lmax, s_base;
    __shared__ uint32_t s_tbl[32];
    __shared__ uint4 s_slow[MODE == 0 ? (VOTE_THREADS / WAVE) * SlowQueue::CAP : 1];

    typedef Chunk<MODE> CH;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int n_slices = a.table.n_slices;
    const int ref_local = (int)(blockIdx.x / n_slices);
    const int ref_ord = a.first_ref + ref_local;
    const int slice = (int)(blockIdx.x % n_slices);
    const uint32_t r = a.ref_idx[ref_ord];
    const uint32_t n_hits = a.hit_count[ref_local];
    const uint4 *hits = reinterpret_cast<const uint4 *>(a.hits_sorted) + (size_t)ref_local * a.hit_stride;

    for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) acc[c] = 0;
    if (tid < 32) s_tbl[tid] = k_alpha_thr[tid];

    const oslamk_slot *tab = a.table.slots + (size_t)slice * a.table.cap;
    const uint32_t mask = a.table.cap - 1, shift = a.table.shift;
    const uint32_t m_base = (uint32_t)slice * OSLAMK_SLICE;   /* first model reference of the slice */
    unsigned long long my_hits = 0, my_votes = 0;
    SlowQueue sq;
    sq.q = s_slow + (MODE == 0 ? wid * SlowQueue::CAP : 0);
    sq.n = 0;
    __syncthreads();

    for (uint32_t base = 0; base < n_hits; base += VOTE_THREADS) {
        const uint32_t h = base + tid;
        const bool have = h < n_hits;
        uint32_t key = 0, cs2v = 0;
        bool fv = false;
        float vyv = 0.0f, vzv = 0.0f;
        if (have) {
            const uint4 rec = hits[h];
            uint32_t c = rec.w + 15u * PC_Q17_ONE;
            key = rec.x;
            vyv = __builtin_bit_cast(float, rec.y);
            vzv = __builtin_bit_cast(float, rec.z);
            fv = rec.w == PC_Q17_FORCE;
            cs2v = c >= PC_Q17_TURN ? c - PC_Q17_TURN : c;
            if (slice == 0) my_hits += 1;
        }
        /* runs of equal keys inside this wave's 64 hits: the first lane of a run probes the
         * slice table; the run ends where the next one starts (or at the last hit) */
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, WAVE);
        const bool is_head = have && (lane == 0 || key != prev);
        const unsigned long long heads = __ballot(is_head), haves = __ballot(have);
        uint32_t start = 0, len = 0, bflag = 0;
        int run = 0;
        if (is_head) {
            const unsigned long long later = heads & ~((2ull << lane) - 1ull);   /* heads above this lane */
            const int end = later ? __ffsll((long long)later) - 1 : __popcll(haves);
            run = end - lane;
            uint32_t slot = slot_of(key, shift);
            for (uint32_t probe = 0; probe <= mask; probe++) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(&tab[slot]);
                if (sv.x == key) {
                    start = sv.y;
                    len = sv.z;
                    bflag = sv.w >> 31;                       /* the bucket holds an entry with the marker */
                    my_votes += (unsigned long long)len * (unsigned)run;
                    break;
                }
                if (sv.x == 0) break;
                slot = (slot + 1) & mask;
            }
        }
        /* wave-cooperative sweep: all 64 lanes stream one bucket at a time in chunks of 256
         * entries, every hit of the run votes with the chunk, and the next chunk's load
         * (possibly of the next bucket) is in flight meanwhile.  Generator state is wave-uniform. */
        unsigned long long todo = __ballot(len > 0);
        ChunkDesc g;                       /* the next chunk to hand out */
        g.valid = todo != 0;
        auto open_bucket = [&](int l) {
            g.st = readlane_u(start, l);
            g.ln = readlane_u(len, l);
            g.head = l;
            g.run = (int)readlane_u((uint32_t)run, l);
            g.bforced = readlane_u(bflag, l) != 0;
            g.off = 0;
        };
        if (g.valid) open_bucket(__ffsll((long long)todo) - 1);
        auto next_chunk = [&]() -> ChunkDesc {
            const ChunkDesc out = g;
            if (g.valid) {
                g.off += 4 * WAVE;
                if (g.off >= g.ln) {
                    todo &= todo - 1;
                    if (todo) open_bucket(__ffsll((long long)todo) - 1);
                    else g.valid = false;
                }
            }
            return out;
        };
        /* VOTE_PIPE chunks are in flight ahead of the one being voted */
        ChunkDesc d[VOTE_PIPE];
        CH c[VOTE_PIPE];
#pragma unroll
        for (int i = 0; i < VOTE_PIPE; i++) {
            d[i] = next_chunk();
            if (d[i].valid) c[i].load(a, d[i], lane);
        }
        while (d[0].valid) {
            const CH cur = c[0];
            const ChunkDesc dc = d[0];
#pragma unroll
            for (int i = 0; i + 1 < VOTE_PIPE; i++) {
                c[i] = c[i + 1];
                d[i] = d[i + 1];
            }
            d[VOTE_PIPE - 1] = next_chunk();
            if (d[VOTE_PIPE - 1].valid) c[VOTE_PIPE - 1].load(a, d[VOTE_PIPE - 1], lane);
            cur.vote(a, acc, s_tbl, sq, dc, lane, cs2v, fv, vyv, vzv);
        }
    }
    if (MODE == 0) sq.flush(a, acc, s_tbl, lane);
    __syncthreads();

    /* ---- peak extraction: local max, non-empty cells, emission ---- */
    uint32_t lmax = 0, nz = 0;
    for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) {
        const uint32_t v = acc[c];
        lmax = v > lmax ? v : lmax;
        nz += (v != 0);
    }
    lmax = wave_max_u32(lmax);
    nz = wave_sum_u32(nz);
    my_hits = wave_sum_u64(my_hits);
    my_votes = wave_sum_u64(my_votes);
    if (lane == 0) {
        s_wave[wid] = lmax;
        s_wave2[wid] = nz;
        s_wave64[0][wid] = my_hits;
        s_wave64[1][wid] = my_votes;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0, n = 0;
        unsigned long long h = 0, v = 0;
        for (int w = 0; w < VOTE_THREADS / WAVE; w++) {
            m = s_wave[w] > m ? s_wave[w] : m;
            n += s_wave2[w];
            h += s_wave64[0][w];
            v += s_wave64[1][w];
        }
        uint32_t g = a.fixed_gmax;
        if (g == 0) {
            const uint32_t old = atomicMax(&a.counters->gmax, m);
            g = old > m ? old : m;
        }
        if (h) atomicAdd(&a.counters->hits, h);
        if (v) atomicAdd(&a.counters->votes, v);
        if (n) atomicAdd(&a.counters->nonzero_cells, (unsigned long long)n);
        s_g = g;
        s_lmax = m;
    }
    __syncthreads();

    if (a.acc_dump && ref_ord == a.dump_ref) {
        uint32_t *dst = a.acc_dump + (size_t)m_base * OSLAMK_NBIN;
        for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) dst[c] = acc[c];
    }

    /* cells with count > thresh * g (model.cu:164-167; g <= final maximum, so
     * this is a superset that the host filters with the final maximum) */
    const float bound = a.thresh * (float)s_g;
    if ((float)s_lmax > bound) {                 /* workgroup-uniform */
        uint32_t cnt = 0;
        for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) cnt += ((float)acc[c] > bound);
        /* exclusive scan of cnt over the workgroup */
        uint32_t incl = cnt;
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += up;
        }
        if (lane == WAVE - 1) s_wave[wid] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int w = 0; w < VOTE_THREADS / WAVE; w++) {
                const uint32_t v = s_wave[w];
                s_wave[w] = run;
                run += v;
            }
            s_base = atomicAdd(&a.counters->out_count, run);
        }
        __syncthreads();
        uint32_t pos = s_base + s_wave[wid] + incl - cnt;
        for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) {
            const uint32_t v = acc[c];
            if ((float)v > bound) {
                if (pos < a.out_cap) {
                    oslamk_cell cell;
                    const uint32_t m_r = m_base + (uint32_t)(c / OSLAMK_NBIN);
                    cell.code = ((unsigned long long)r << 32) | (unsigned long long)((m_r << 6) |
                                                                 (uint32_t)(c % OSLAMK_NBIN));
                    cell.count = v;
                    cell.pad = 0;
                    a.out[pos] = cell;
                }
                pos++;
            }
        }
    }
}

/* --------------------------------------------------------------------------
 * pose clustering scores (rot_clustering_kernel, kernel.cu:702-763, without the in-place
 * translation averaging): one thread per pose scans its 26 neighbour cells in the fixed
 * (dx,dy,dz) order and sums the weighted votes of compatible poses.  Every pose's float sum
 * has its own fixed order, so the result equals the host loop of oslam_pose.c bit for bit.
 * sorted[] = (cell hash, pose index) ascending; a hash of 0 is never searched (kernel.cu:727).
 * ------------------------------------------------------------------------*/
__device__ __forceinline__ uint32_t fnv_cell3(int cx, int cy, int cz)
{
    uint32_t h = pm_fnv1a_word(PM_FNV_BASIS, (uint32_t)cx);
    h = pm_fnv1a_word(h, (uint32_t)cy);
    return pm_fnv1a_word(h, (uint32_t)cz);
}

/* One wave per pose.  The 64 lanes test 64 candidates of a neighbour cell at once (the
 * expensive part); the compatible ones are then added by walking the ballot mask in ascending
 * order, i.e. exactly the sequential float sum of the reference loop.  sq/st/sw are the pose
 * quaternions, translations and weighted votes permuted into sorted (cell hash, pose index)
 * order, so consecutive lanes read consecutive memory. */
__global__ __launch_bounds__(64) void k_cluster_scores(int n, const float *trans, const float *quat,
                                                       const int *cell, const uint32_t *shash,
                                                       const float4 *sq, const float *st, const float *sw,
                                                       float d_dist, int use_l1, float *score)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const float rot_thresh = 2 * PM_D_ANGLE, rot_thresh_sq = rot_thresh * rot_thresh;
    const float q0 = quat[4 * i], q1 = quat[4 * i + 1], q2 = quat[4 * i + 2], q3 = quat[4 * i + 3];
    const float tx = trans[3 * i], ty = trans[3 * i + 1], tz = trans[3 * i + 2];
    const int cx = cell[3 * i], cy = cell[3 * i + 1], cz = cell[3 * i + 2];
    float votes = 1;                                             /* kernel.cu:722 */
    for (int dx = -1; dx < 2; dx++)
        for (int dy = -1; dy < 2; dy++)
            for (int dz = -1; dz < 2; dz++) {
                if (dx == 0 && dy == 0 && dz == 0) continue;     /* kernel.cu:684-689 */
                const uint32_t h = fnv_cell3(cx + dx, cy + dy, cz + dz);
                if (h == 0) continue;
                int lo = 0, hi = n;
                while (lo < hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if (shash[mid] < h) lo = mid + 1; else hi = mid;
                }
                for (int base = lo; base < n; base += WAVE) {
                    const int j = base + lane;
                    bool ok = false;
                    float w = 0.0f;
                    const bool in_cell = j < n && shash[j] == h;
                    if (in_cell) {
                        const float4 qo = sq[j];
                        const float qd = fabsf(8 * (1 - (q0 * qo.x + q1 * qo.y + q2 * qo.z + q3 * qo.w)));
                        ok = qd < rot_thresh_sq;
                        if (ok && !use_l1) {
                            const float ex = tx - st[3 * j], ey = ty - st[3 * j + 1], ez = tz - st[3 * j + 2];
                            ok = pm_sqrtf(ex * ex + ey * ey + ez * ez) < d_dist;
                        }
                        w = sw[j];
                    }
                    unsigned long long m = __ballot(ok);
                    while (m) {                                  /* ascending j: the reference's order */
                        const int b = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        votes += readlane_f(w, b);
                    }
                    if (__ballot(in_cell) != ~0ull) break;       /* the cell's run ended in this step */
                }
            }
    if (lane == 0) score[i] = votes;
}

/* --------------------------------------------------------------------------
 * device self-test of the float path
 * ------------------------------------------------------------------------*/
__global__ void k_selftest(const float *x, const float *y, const float *x2, size_t n, float *out_acos,
                           float *out_atan2, uint32_t *out_quant, uint32_t *out_bin)
{
    __shared__ uint32_t s_tbl[32];
    if (threadIdx.x < 32) s_tbl[threadIdx.x] = k_alpha_thr[threadIdx.x];
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_acos[i] = pm_acosf(x[i]);
    out_atan2[i] = pm_atan2f(y[i], x2[i]);
    out_quant[i] = pc_quant_bits(pm_fabsf(y[i]) * 7.0f, 0.0371f + pm_fabsf(x[i]) * 0.01f,
                                 1.0f / (0.0371f + pm_fabsf(x[i]) * 0.01f));
    out_bin[i] = pc_alpha_bin_table(y[i], x2[i], x[i], y[i] - x2[i], s_tbl);
}

/* --------------------------------------------------------------------------
 * launchers
 * ------------------------------------------------------------------------*/
extern "C" {

int oslamk_row_keys(oslamk_cloud c, int ref, float d_dist, float inv_d_dist, uint32_t *keys_out,
                    void *stream)
{
    hipLaunchKernelGGL(k_row_keys, dim3((c.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, ref,
                       d_dist, inv_d_dist, keys_out);
    return (int)hipGetLastError();
}

int oslamk_model_count(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                       uint32_t *n_unique, uint32_t *overflow, void *stream)
{
    hipLaunchKernelGGL(k_model_count, dim3((c.n + 255) / 256, c.n), dim3(256), 0, (hipStream_t)stream,
                       c, d_dist, inv_d_dist, t, n_unique, overflow);
    return (int)hipGetLastError();
}

int oslamk_table_scan(oslamk_table t, uint32_t *total_out, void *stream)
{
    hipLaunchKernelGGL(k_table_scan, dim3(1), dim3(1024), 0, (hipStream_t)stream, t, total_out);
    return (int)hipGetLastError();
}

int oslamk_union_build(oslamk_table t, uint32_t *n_keys, uint32_t *overflow, void *stream)
{
    size_t total = (size_t)t.n_slices * t.cap;
    hipLaunchKernelGGL(k_union_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, t, n_keys, overflow);
    return (int)hipGetLastError();
}

int oslamk_model_fill(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                      const float *tmg, oslamk_entries ent, void *stream)
{
    hipLaunchKernelGGL(k_model_fill, dim3((c.n + 255) / 256, c.n), dim3(256), 0, (hipStream_t)stream,
                       c, d_dist, inv_d_dist, t, tmg, ent);
    return (int)hipGetLastError();
}

int oslamk_reach_build(oslamk_table t, float d_dist, void *stream)
{
    hipLaunchKernelGGL(k_reach_build, dim3(OSLAMK_REACH_BINS), dim3(256), 0, (hipStream_t)stream, t, d_dist);
    return (int)hipGetLastError();
}

int oslamk_scene_hits(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    dim3 grid((unsigned)((a->scene.n + KEY_TILE - 1) / KEY_TILE), (unsigned)a->n_launch);
    hipLaunchKernelGGL(k_scene_hits, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

int oslamk_sort_hits(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    hipLaunchKernelGGL(k_sort_hits, dim3((unsigned)a->n_launch), dim3(1024), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

int oslamk_vote(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    dim3 grid((unsigned)((size_t)a->n_launch * a->table.n_slices));
    if (a->mode == 0)
        hipLaunchKernelGGL(k_vote<0>, grid, dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    else
        hipLaunchKernelGGL(k_vote<1>, grid, dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

int oslamk_cluster_scores(int n, const float *trans, const float *quat, const int *cell, const uint32_t *shash,
                          const float *sq, const float *st, const float *sw, float d_dist, int use_l1,
                          float *score, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_cluster_scores, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, n, trans, quat, cell,
                       shash, reinterpret_cast<const float4 *>(sq), st, sw, d_dist, use_l1, score);
    return (int)hipGetLastError();
}

int oslamk_selftest(const float *x, const float *y, const float *x2, size_t n, float *out_acos,
                    float *out_atan2, uint32_t *out_quant, uint32_t *out_bin, void *stream)
{
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, x2, n, out_acos, out_atan2, out_quant, out_bin);
    return (int)hipGetLastError();
}

} /* extern "C" */
