/*
 * oslam_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the PPF registration
 * path.  Wave = 64 lanes throughout.  Built with -ffp-contract=off: the float
 * sequences of ppf_core.h must round exactly as on the host.
 *
 * What the reference does with N^2-sized arrays and Thrust sorts
 * (pcl/alignment/src/cuda/{scene,model}.cu) is fused here:
 *   model build : pair key -> per-slice open-addressing table -> bucketed
 *                 4-byte pair entries (two counting passes, no sort); union table
 *                 of all keys, and per slice the bucket of every union slot; bitset
 *                 of the distance bins that can reach a key; key map (distance bin,
 *                 three angle bins) -> union slot; entries ordered inside each
 *                 bucket for the LDS banks;
 *   scene count : per reference point, the pairs whose distance bin can reach a
 *                 model key: sizes the hit lists by demand;
 *   scene keys  : per (8 reference points, tile of scene points): distance bin of
 *                 every pair, unreachable bins dropped, the rest compacted in LDS;
 *                 the pair's feature as bins -> key map (one load, no hash, no
 *                 probing) -> per-reference hit list {union slot} + {theta_v, i}
 *                 written by wave-aggregated appends;
 *   hit sort    : per reference point, hits ordered by union slot (LDS radix
 *                 sort) and the list of runs of equal keys;
 *   voting      : one workgroup per (scene reference point, model slice of 2046
 *                 points): a run finds its bucket with one load (no probing); very
 *                 long items are cut into units dealt to the 16 waves, the rest is
 *                 handed out dynamically a few runs at a time; a wave streams a
 *                 bucket once for all hits that share it (16 bytes = 4 entries per
 *                 lane, four steps' loads in flight) -> integer alpha bin -> LDS
 *                 accumulator [1023 rows + sink][31 words], two 16-bit counters per
 *                 word; votes near a bin edge are cast, noted and re-checked with
 *                 the reference's float sequence; in-kernel peak extraction;
 *   clustering  : scores of the candidate poses (one wave per pose).
 */
#include <hip/hip_runtime.h>

#include "oslam_kernels.h"
#include "ppf_core.h"

#define WAVE 64
#ifndef VOTE_THREADS
#define VOTE_THREADS 1024
#endif
/* The accumulator of a vote workgroup in LDS: OSLAMK_ROWS rows of ACC_STRIDE words (bins 0..29 + one unused).
 * The stride is odd on purpose: the LDS bank of a vote is (row * 31 + bin) mod 32 = (bin - row) mod 32, so votes
 * of one instruction that fall into the same bin of different rows do not meet on a bank, and k_bucket_spread
 * can order a bucket so that the 32 lanes the LDS serves together land on 32 different banks whatever the
 * hit's angle is (a stride of 32 would make the bank the bin alone: 30 banks, and as crowded as the model's
 * angles are). */
#ifndef ACC_STRIDE
#define ACC_STRIDE 31
#endif
#define ACC_CELLS (OSLAMK_ROWS * ACC_STRIDE)
#define ACC_REAL_CELLS (1023 * ACC_STRIDE)       /* counter words without the sink row; each holds two 16-bit counters */

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

/* table.uids: the keys of the union table numbered 0 .. n-1 (any order).  One thread per slot. */
__global__ void k_union_ids(oslamk_table t, uint32_t *counter)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < t.ucap && t.ukeys[slot] != 0u) t.uids[slot] = atomicAdd(counter, 1u);
}

/* table.kmap[k1][combo] = the number (table.uids) of the key that distance bin k1 and the angle bins
 * `combo` hash to (pc_key_of_bins), or OSLAMK_KMAP_NONE: the scene-key kernel then needs neither the hash nor
 * a probe sequence.  One workgroup per distance bin k1 < table.kmap_bins. */
__global__ __launch_bounds__(256) void k_kmap_build(oslamk_table t, float d_dist)
{
    const uint32_t k1 = blockIdx.x, mask = t.ucap - 1;
    for (uint32_t combo = threadIdx.x; combo < PC_ANGLE_COMBOS; combo += 256) {
        const uint32_t key = pc_key_of_bins(k1, combo, d_dist);
        uint32_t found = OSLAMK_KMAP_NONE;
        if (key != 0) {
            uint32_t slot = slot_of(key, t.ushift);
            for (uint32_t probe = 0; probe <= mask; probe++) {
                const uint32_t k = t.ukeys[slot];
                if (k == key) { found = t.uids[slot]; break; }
                if (k == 0) break;
                slot = (slot + 1) & mask;
            }
        }
        t.kmap[(size_t)k1 * PC_ANGLE_COMBOS + combo] = found;
    }
}

/* table.uinfo[slice][slot of the key in the union table] = the key's bucket in that slice: lets the vote
 * kernel go from a hit to its bucket with one load instead of a probe sequence.  One thread per slot of
 * the slice tables; uinfo is zeroed by the host (len 0 = the slice has no pair with that key). */
__global__ void k_uinfo_build(oslamk_table t)
{
    const size_t total = (size_t)t.n_slices * t.cap;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const oslamk_slot sl = t.slots[idx];
    if (sl.key == 0) return;
    const uint32_t mask = t.ucap - 1;
    uint32_t slot = slot_of(sl.key, t.ushift);
    for (uint32_t probe = 0; probe <= mask; probe++) {
        if (t.ukeys[slot] == sl.key) {
            oslamk_uinfo ui;
            ui.start = sl.start;
            ui.len = sl.len | (sl.cur & 0x80000000u);      /* bit 31 of the fill cursor: marker entry in the bucket */
            t.uinfo[(idx / t.cap) * (size_t)t.uinfo_stride + t.uids[slot]] = ui;
            return;
        }
        slot = (slot + 1) & mask;
    }
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
        const uint32_t th = pc_angle_t22(uy, uz);
        /* a marker forces the whole bucket through the exact path: flagged in bit 31 of the cursor */
        if (th == PC_T22_FORCE) atomicOr(&tab[slot].cur, 0x80000000u);
        ent.e4[e] = pc_entry_word(th == PC_T22_FORCE ? 0u : th, pc_row11((uint32_t)(m_r - slice * OSLAMK_SLICE)));
    }
    ent.mi[e] = (uint16_t)i;
    if (ent.uv) {
        oslamk_uv en;
        en.uy = uy;
        en.uz = uz;
        ent.uv[e] = en;
    }
}

/* pass 3: order inside every bucket.  A vote instruction adds 64 lanes' entries into acc[row][bin]; the LDS
 * serves 32 lanes together, one cycle per distinct address on the busiest bank (two are free: the instruction
 * takes four cycles to hand over its operands anyway).  The fill pass leaves the entries in arrival order:
 * 32 random banks collide 3-4 deep (7 LDS cycles per instruction instead of 4, tools/micro/lds_atomic_bench.hip).
 * Here every segment of 4096 positions (16 chunks) of a bucket is sorted by spread_key -- the quantity that
 * decides an entry's bank for every hit angle at once -- and dealt out so that each such group of 32 (same
 * chunk, same register j, same half of the wave) takes every G-th entry of the sorted order: keys about one
 * bank apart, mostly distinct banks.  Votes commute, so the order inside a bucket is free (model.cu:95-171
 * sorts them anyway).  One workgroup per slot. */
#define SPREAD_SEG 4096
#define SPREAD_THREADS 256
/* What the entries of a bucket are ordered by: the bank a vote lands on is floor(s - kappa) mod 32 with
 * kappa = (theta_u in bins - ACC_STRIDE * row) mod 32 and s the hit's angle in bins (but for the wrap of the bin
 * at 30, which moves a bank by 2), so lanes whose kappa are one apart never collide, whatever s is.  In units
 * of 2^-16 bank. */
__device__ __forceinline__ uint32_t spread_key(uint32_t entry_word)
{
    const uint32_t u = ((entry_word >> PC_ROW_BITS) * 30u) >> 5;              /* theta_u21 * 30 / 2^21 bins, << 16 */
    const uint32_t row = entry_word & PC_ROW10_MASK;
    return (u - ((row * ACC_STRIDE) << 16)) & ((32u << 16) - 1u);
}
/* how many positions p' < n of a chunk image (position = 4*lane + j) come before position p in
 * the dealing order (lane & 31, j, lane >> 5), for a chunk that holds n entries */
__device__ __forceinline__ uint32_t spread_rank_in_chunk(uint32_t p, uint32_t n)
{
    const uint32_t lane = p >> 2, j = p & 3u, h = lane & 31u, g = lane >> 5;
    uint32_t r = 0;
    for (uint32_t h2 = 0; h2 < h; h2++) {
        const uint32_t a0 = 4u * h2, a1 = 4u * (h2 + 32u);
        r += (n > a0 ? (n - a0 < 4u ? n - a0 : 4u) : 0u) + (n > a1 ? (n - a1 < 4u ? n - a1 : 4u) : 0u);
    }
    for (uint32_t j2 = 0; j2 < 4; j2++)
        for (uint32_t g2 = 0; g2 < 2; g2++) {
            if (j2 > j || (j2 == j && g2 >= g)) continue;
            r += (4u * (32u * g2 + h) + j2) < n;
        }
    return r;
}
__global__ __launch_bounds__(SPREAD_THREADS) void k_bucket_spread(oslamk_table t, oslamk_entries ent)
{
    __shared__ unsigned long long key[SPREAD_SEG];     /* theta_u << 32 | index in the segment */
    __shared__ uint32_t s_e4[SPREAD_SEG];
    __shared__ oslamk_uv s_uv[SPREAD_SEG];
    __shared__ uint16_t s_mi[SPREAD_SEG];
    const oslamk_slot sl = t.slots[blockIdx.x];
    if (sl.key == 0 || sl.len < 2) return;
    const int tid = threadIdx.x;
    for (uint32_t seg = 0; seg < sl.len; seg += SPREAD_SEG) {
        const uint32_t n = sl.len - seg < SPREAD_SEG ? sl.len - seg : SPREAD_SEG;
        const size_t base = (size_t)sl.start + seg;
        uint32_t P = 2;
        while (P < n) P <<= 1;
        for (uint32_t i = tid; i < P; i += SPREAD_THREADS) {
            if (i < n) {
                const uint32_t w = ent.e4[base + i];
                s_e4[i] = w;
                if (ent.uv) s_uv[i] = ent.uv[base + i];
                s_mi[i] = ent.mi[base + i];
                key[i] = ((unsigned long long)spread_key(w) << 32) | i;
            } else {
                key[i] = ~0ull;
            }
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t x = tid; x < P / 2; x += SPREAD_THREADS) {
                    const uint32_t lo = ((x & ~(j - 1)) << 1) | (x & (j - 1)), hi = lo | j;
                    const unsigned long long a = key[lo], b = key[hi];
                    const bool up = (lo & k) == 0;
                    if ((a > b) == up) { key[lo] = b; key[hi] = a; }
                }
                __syncthreads();
            }
        }
        /* position p of the segment takes the entry whose sorted rank is the number of positions
         * dealt before p: (lane & 31) first, then chunk, then (j, half) */
        const uint32_t n_full = n >> 8, n_last = n & 255u;
        for (uint32_t p = tid; p < n; p += SPREAD_THREADS) {
            const uint32_t c = p >> 8, q = p & 255u, h = (q >> 2) & 31u;
            const uint32_t nc = c < n_full ? 256u : n_last;
            uint32_t rank = 8u * h * n_full + spread_rank_in_chunk(4u * h, n_last);
            rank += 8u * (c < n_full ? c : n_full);
            rank += spread_rank_in_chunk(q, nc) - spread_rank_in_chunk(4u * h, nc);
            const uint32_t src = (uint32_t)key[rank];
            ent.e4[base + p] = s_e4[src];
            if (ent.uv) ent.uv[base + p] = s_uv[src];
            ent.mi[base + p] = s_mi[src];
        }
        __syncthreads();
    }
}

/* pass 4 (exact mode): every bucket once more, in segments of OSLAMK_PSEG entries ordered by P(word) = (word * 30)
 * mod 2^32 -- the position inside its bin that the entry's vote takes, up to the hit's constant (oslamk_entries).
 * ent.pw gets the words, ent.puv their uv (what a re-evaluation reads).  One workgroup per slot. */
__global__ __launch_bounds__(SPREAD_THREADS) void k_bucket_psort(oslamk_table t, oslamk_entries ent)
{
    __shared__ unsigned long long key[OSLAMK_PSEG];    /* P << 32 | index in the segment */
    __shared__ uint32_t s_e4[OSLAMK_PSEG];
    __shared__ oslamk_uv s_uv[OSLAMK_PSEG];
    const oslamk_slot sl = t.slots[blockIdx.x];
    if (sl.key == 0 || sl.len == 0) return;
    const int tid = threadIdx.x;
    for (uint32_t seg = 0; seg < sl.len; seg += OSLAMK_PSEG) {
        const uint32_t n = sl.len - seg < OSLAMK_PSEG ? sl.len - seg : OSLAMK_PSEG;
        const size_t base = (size_t)sl.start + seg;
        uint32_t P = 2;
        while (P < n) P <<= 1;
        for (uint32_t i = tid; i < P; i += SPREAD_THREADS) {
            if (i < n) {
                const uint32_t w = ent.e4[base + i];
                s_e4[i] = w;
                s_uv[i] = ent.uv[base + i];
                key[i] = ((unsigned long long)(w * 30u) << 32) | i;
            } else {
                key[i] = ~0ull;
            }
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t x = tid; x < P / 2; x += SPREAD_THREADS) {
                    const uint32_t lo = ((x & ~(j - 1)) << 1) | (x & (j - 1)), hi = lo | j;
                    const unsigned long long a = key[lo], b = key[hi];
                    const bool up = (lo & k) == 0;
                    if ((a > b) == up) { key[lo] = b; key[hi] = a; }
                }
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < n; i += SPREAD_THREADS) {
            const uint32_t src = (uint32_t)key[i];
            ent.pw[base + i] = s_e4[src];
            ent.puv[base + i] = s_uv[src];
        }
        /* the directory: first position whose cell (the top log2 K bits of P) is >= k */
        const uint32_t K = OSLAMK_PDIR_CELLS(n);
        for (uint32_t k = tid; k <= K; k += SPREAD_THREADS) {
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t cell = (uint32_t)(((key[mid] >> 32) * (unsigned long long)K) >> 32);
                if (cell < k) lo = mid + 1; else hi = mid;
            }
            ent.pdir[base + k] = (uint16_t)lo;
        }
        __syncthreads();
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
/* bin and position inside the bin of one vote, from tm8 = (hit base - 4*theta_u) << 8 (the 24-bit
 * difference in the upper 24 bits of the word: the left shift drops what lies above it):
 * tm8 * 30 = tm24 * 7680 = bin * 2^32 + position * 2^32, one v_mad_u64_u32 (4.3 cycles per
 * wave-instruction, like each of the two 24-bit multiplies it replaces).  The empty asm keeps the
 * compiler from fusing (bin << 2) with the low word into a 64-bit shift. */
__device__ __forceinline__ void vote_product(uint32_t tm8, uint32_t &bin, uint32_t &pos)
{
    const unsigned long long p = (unsigned long long)tm8 * 30ull;
    bin = (uint32_t)(p >> 32);
    pos = (uint32_t)p;
    asm("" : "+v"(bin));
}
/* a value every lane holds, moved to scalar registers */
__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni_u64(unsigned long long v)
{
    return ((unsigned long long)uni_u32((uint32_t)(v >> 32)) << 32) | uni_u32((uint32_t)v);
}

/* ---- the vote of one (model pair entry, scene hit) ---------------------------
 * A wave holds a chunk of 256 model-pair entries in registers: lane l has entries 4l .. 4l+3
 * of the chunk (one 16-byte load).  Every hit of the run votes with the chunk:
 *     tm  = hit's base (theta_v + half a turn + margin, scaled; one SGPR) - the entry word as stored
 *     {bin, position} = v_mad_u64_u32(tm, 30)
 * Only positions within the margin of a bin edge (0.05 % of votes) are re-evaluated with the
 * reference's float sequence via pc_alpha_bin_table (ppf_core.h), so the bins are the
 * reference's.  The code is straight-line: lanes past the end of the bucket are masked out of
 * the atomics (EXEC), padding entries vote into the accumulator's sink row. */
#define ACC_TRASH_WORDS 96              /* slack behind the accumulator (16-byte zeroing, alignment of what follows) */

/* what the re-evaluation of a vote reads: the model's entries in the order of oslamk_entries.pw / .puv, the
 * reference point's hit list, the scene cloud and the rows y,z of T_s_g (kernel.cu:334-336) */
struct SlowCtx {
    const uint32_t *pw;
    const oslamk_uv *puv;
    const oslamk_pay *hits;
    const float *px, *py, *pz;
    const float *rows;
    uint32_t inc_lo, inc_hi;           /* what a vote adds to its counter word: by the half the model point owns */
    uint32_t dropped;                  /* re-evaluated votes that fell into no bin (alpha not a number) */
};

/* The second instantiation of ppf_math_atan.inc (see there): pm_atan2f_cold / pc_alpha_bin_exact_cold with every
 * literal materialised where it is used.  The re-evaluation of a vote takes the table path (pc_alpha_bin_table_main)
 * and this formula only for inputs the table does not cover -- zero or non-finite cross and dot products. */
#define PM_FN(n) n##_cold
#define PM_KU(bits) ({ uint32_t pinned_; asm volatile("v_mov_b32 %0, %1" : "=v"(pinned_) : "i"(bits)); pinned_; })
#define PM_KF(bits) __builtin_bit_cast(float, PM_KU(bits))
#undef PM_HD
#define PM_HD __device__ static inline
#include "ppf_math_atan.inc"
#undef PM_FN
#undef PM_KU
#undef PM_KF
__device__ __forceinline__ unsigned alpha_bin_reeval(float uy, float uz, float vy, float vz, const uint32_t *tbl)
{
    const unsigned b = pc_alpha_bin_table_main(uy, uz, vy, vz, tbl);
    return b != PC_ALPHA_OUTSIDE ? b : pc_alpha_bin_exact_cold(uy, uz, vy, vz);
}

/* LDS pointers of the out-of-line paths; their context lives in LDS too (a kernel's stack would be scratch memory) */
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) SlowCtx lds_ctx;

/* Per-wave queue (LDS) of votes to re-evaluate with pc_alpha_bin_table: {entry index (pw order), hit index};
 * their operands are a dependent gather, so they are evaluated 64 at a time by slow_queue_flush().
 * q: the wave's SLOW_CAP places; n: how many are taken (wave-uniform). */
#define SLOW_CAP 96
#define SLOW_VERIFY 0x80000000u        /* in an item's entry index: the vote has been cast into its quantised bin */

__device__ __forceinline__ void slow_queue_flush(lds_ctx *sc, lds_u32 *acc, lds_u32 *tbl, lds_u64 *q, uint32_t n, int lane)
{
    const uint32_t *t = (const uint32_t *)tbl;
    for (uint32_t base = 0; base < n; base += WAVE) {
        if (base + lane < n) {
            const unsigned long long it = q[base + lane];
            const uint32_t entry = (uint32_t)it & ~SLOW_VERIFY;
            const oslamk_pay hp = sc->hits[(uint32_t)(it >> 32)];
            const uint32_t i = hp.idx;
            const float x = sc->px[i], y = sc->py[i], z = sc->pz[i];
            const float vy = pc_row_dot(sc->rows, x, y, z);        /* as k_scene_hits computed them */
            const float vz = pc_row_dot(sc->rows + 4, x, y, z);
            const uint32_t ew = sc->pw[entry];
            const uint32_t mr = ew & PC_ROW10_MASK;
            const uint32_t inc = (ew >> PC_ROW_HALF_BIT) & 1u ? sc->inc_hi : sc->inc_lo;
            const float2 uv = *reinterpret_cast<const float2 *>(&sc->puv[entry]);
            const unsigned bin = alpha_bin_reeval(uv.x, uv.y, vy, vz, t);
            if (!((uint32_t)it & SLOW_VERIFY)) {
                /* a vote that has not been cast (an item with a marker) */
                if (bin < OSLAMK_NBIN) atomicAdd((uint32_t *)&acc[mr * ACC_STRIDE + bin], inc);
                else atomicAdd((uint32_t *)&sc->dropped, 1u);
            } else {
                /* a vote that has been cast into its quantised bin and lies within the margin of a bin edge: moved if
                 * the reference's float sequence puts it into another bin (a counter word is only ever added to and
                 * subtracted from, so a carry between its two halves that a misplaced vote caused is undone with it,
                 * whichever of the two comes first) */
                uint32_t qbin, pos;
                vote_product(pc_vote_base_t32(hp.theta_t22) - ew, qbin, pos);
                if (bin != qbin) {
                    atomicSub((uint32_t *)&acc[mr * ACC_STRIDE + qbin], inc);
                    if (bin < OSLAMK_NBIN) atomicAdd((uint32_t *)&acc[mr * ACC_STRIDE + bin], inc);
                    else atomicAdd((uint32_t *)&sc->dropped, 1u);
                }
            }
        }
    }
}

/* appends the votes of `mask`'s lanes (entry, hit) to the wave's queue, emptying it first when they would not fit */
__device__ __forceinline__ uint32_t slow_push(lds_ctx *sc, lds_u32 *acc, lds_u32 *tbl, lds_u64 *q, uint32_t n,
                                              unsigned long long mask, uint32_t entry, uint32_t hit, int lane)
{
    if (n > SLOW_CAP - WAVE) {
        slow_queue_flush(sc, acc, tbl, q, n, lane);
        n = 0;
    }
    if ((mask >> lane) & 1ull)
        q[n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned long long)entry | ((unsigned long long)hit << 32);
    return n + (uint32_t)__popcll(mask);
}

/* One step of a wave (all fields wave-uniform): one chunk of a bucket voted by the hits
 * i0 .. i1-1 of a run (at most 64 hits, one per lane). */
struct VoteStep {
    uint32_t e0;                       /* index of the chunk's first entry */
    uint32_t left;                     /* entries from the chunk start to the bucket end (>= 1), capped at 256 */
    uint32_t h0;                       /* index of the run's first hit in the reference point's sorted list */
    uint32_t R;                        /* hits of the run */
    uint32_t i0, i1;                   /* the hits of the run that vote in this step */
    bool valid;
};

/* An item whose every vote is re-evaluated with the reference's float sequence -- its bucket holds an
 * entry with the marker, or one of its hits carries it (degenerate geometry: a second point on the
 * reference point's normal, a non-finite coordinate): every (entry, hit) pair goes through the queue.
 * Rare and slow on purpose; used after the vote loop, never inside it (inlined there: a call anywhere in the
 * kernel costs callee-saved registers and spills). */
__device__ __forceinline__ uint32_t forced_item(lds_ctx *sc, lds_u32 *acc, lds_u32 *tbl, lds_u64 *q, uint32_t n, uint32_t st,
                                             uint32_t ln, uint32_t h0, uint32_t R, int lane)
{
    for (uint32_t e = 0; e < ln; e += WAVE) {
        const unsigned long long nm = __ballot(e + (uint32_t)lane < ln);
        for (uint32_t i = 0; i < R; i++) n = slow_push(sc, acc, tbl, q, n, nm, st + e + (uint32_t)lane, h0 + i, lane);
    }
    return n;
}

/* The registers of a step in flight and its votes: lane l holds entries 4l .. 4l+3 of the chunk (one
 * 16-byte load) and theta_v of hit `lane` of the run.  One code path for full and partial chunks: the
 * atomics are issued with EXEC narrowed to the lanes that hold entries, so idle lanes cost no LDS cycles
 * and cause no bank conflicts; the padding behind a bucket votes into the sink row.  Written in asm because
 * the compiler has no way to say this; the workgroup waits for these atomics (lgkmcnt) before it reads the
 * accumulator.  Per hit (256 votes): v_readlane + 4 x (v_sub, v_mad_u64_u32, v_lshl_add) + (exact mode)
 * v_min, v_min3, v_cmp = 16 vector instructions, 60 cycles of a SIMD + 4 LDS atomics + loop control.  Items
 * that carry a marker never come here (forced_item). */
template <int MODE>
struct VoteRegs {
    uint4 v;
    uint32_t th;
    /* Unconditional on purpose: every lane loads, whatever the chunk and the run hold, so that the loads of
     * a group are straight-line code and the compiler can wait for each member's registers alone
     * (vmcnt(N)) instead of draining everything. */
    __device__ __forceinline__ void load(const uint32_t *e4, const oslamk_pay *hits, const VoteStep &d, int lane)
    {
        /* lanes past the end of the bucket / the run re-read its first bytes (no extra traffic); wave-uniform base
         * + 32-bit lane offset, so that the address costs one select per load */
        const uint32_t lane16 = 16u * (uint32_t)lane, lane8 = 8u * (uint32_t)lane;
        v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(e4 + d.e0) + ((uint32_t)lane < (d.left + 3u) >> 2 ? lane16 : 0u));
        th = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(&hits[d.h0].theta_t22) +
                                                 ((uint32_t)lane < d.R ? lane8 : 0u));
    }
    __device__ __forceinline__ void vote(uint32_t *accp, const VoteStep &d, int lane, uint32_t inc_lo, uint32_t inc_hi) const
    {
        lds_u32 *acc = (lds_u32 *)accp;
        /* the entry words as they are: theta_u << 11 | half << 10 | row (pc_entry_word); lanes and words past the
         * end of the bucket are masked out of the atomics (EXEC) or are padding entries that vote into the sink
         * row.  A counter word holds two 16-bit counters: the model points of the lower half of the slice add 1,
         * those of the upper half 0x10000 (inc_lo / inc_hi; one of them is 0 in the rare second and third pass
         * of a workgroup whose 16-bit counters overflowed) */
        const uint32_t wa[4] = {v.x, v.y, v.z, v.w};
        /* pc_vote_base_t32(th) = (th << 10) + a constant, which is kept in a scalar register */
        uint32_t csmv;
        asm("v_lshl_add_u32 %0, %1, 10, %2" : "=v"(csmv) : "v"(th), "s"(pc_vote_base_t32(0u)));
        const uint32_t acc_base = (uint32_t)(uintptr_t)acc;  /* the accumulator's LDS address */
        uint32_t rowb[4], inc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            rowb[j] = acc_base + (wa[j] & PC_ROW10_MASK) * (4u * ACC_STRIDE);     /* LDS address of the entry's row */
            /* inc_lo or inc_hi by the half bit: a bit-field extract and (first pass) one 24-bit multiply-add, in asm
             * because the compiler turns every spelling of it into and + compare + select */
            if (inc_hi == 0x10000u)
                asm("v_bfe_u32 %0, %1, 10, 1\n\tv_mad_u32_u24 %0, %0, %2, 1" : "=&v"(inc[j]) : "v"(wa[j]), "s"(0xffffu));
            else                         /* the wide passes count one half of the slice's model points with 32-bit counters: 1 or 0 */
                inc[j] = ((wa[j] >> PC_ROW_HALF_BIT) & 1u) ^ (inc_lo ? 1u : 0u);
        }
        /* lanes that hold at least one entry of this chunk */
        const unsigned long long live = __ballot((uint32_t)lane < (d.left + 3u) >> 2);
#ifdef VOTE_DIAG_NOLOOP                 /* timing-only build: steps and loads without the votes */
        asm volatile("" ::"v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(csmv), "s"(live));
        return;
#endif
        /* Every vote goes into its quantised bin, in both modes: v_readlane + 4 x (v_sub, v_mad_u64_u32, v_lshl_add) =
         * 13 vector instructions and 4 LDS atomics per hit and chunk.  In exact mode the votes that lie within the
         * margin of a bin edge are found afterwards, by search (vote_body: correct_set), and moved where the
         * reference's float sequence says so. */
        for (uint32_t i = d.i0; i < d.i1; i++) {
            const uint32_t csm = readlane_u(csmv, (int)i);
            uint32_t addr[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t bin, pos;
                vote_product(csm - wa[j], bin, pos);
                addr[j] = rowb[j] + (bin << 2);
            }
#ifdef VOTE_DIAG_NOATOM             /* timing-only build: the vote arithmetic without the LDS atomics */
            asm volatile("" ::"v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]));
            continue;
#endif
            unsigned long long saved;
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b64 exec, %1\n\t"
                         "ds_add_u32 %2, %6\n\t"
                         "ds_add_u32 %3, %7\n\t"
                         "ds_add_u32 %4, %8\n\t"
                         "ds_add_u32 %5, %9\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(live), "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(inc[0]), "v"(inc[1]),
                           "v"(inc[2]), "v"(inc[3])
                         : "memory");
        }
    }
};

/* --------------------------------------------------------------------------
 * scene pair keys -> hit lists (Scene::Scene's key pass, scene.cu:24-55: K1 ppf_kernel + K2
 * ppf_hash_kernel, fused with the lookup of model.cu:96-97)
 * ------------------------------------------------------------------------*/
#define KEY_TILE 4096                  /* scene points per counting workgroup */
#define COUNT_REFS 8                   /* reference points one workgroup handles against its tile */
#define HIT_TILE 1024                  /* scene points per hit-list workgroup (indices inside it fit 16 bits) */

__device__ const uint32_t d_acos_lut[2 * PC_ACOS_CELLS] = {PC_ACOS_LUT_FLAT};

/* The distance bin of a pair, pc_pair_dist_bin's value at a third of its cost: the hardware's approximate
 * square root (1 ulp) gives the quotient to 3e-7 relative, which decides the bin unless the quotient lies
 * within 1e-6 relative of a bin edge (or the distance is tiny, huge or not a number) -- those lanes, a few
 * in a million, take the exact sequence. */
__device__ __forceinline__ int pair_dist_bin_quick(float dx, float dy, float dz, float d_dist, float inv_d_dist)
{
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float q = __builtin_amdgcn_sqrtf(d2) * inv_d_dist;
    const float kf = __builtin_floorf(q);
    const float fr = q - kf, tol = q * 1e-6f;
    int k = (int)kf;
    if (!(d2 > 1e-30f && q < 1048576.0f && fr > tol && 1.0f - fr > tol)) k = pc_pair_dist_bin(dx, dy, dz, d_dist, inv_d_dist);
    return k;
}

/* Can a pair in distance bin k produce a key of the model at all?  Exact: table.reach has a bit for every
 * distance bin that holds a model key, FNV collisions included (`reach` = the workgroup's LDS copy of its first
 * reach_words words, the rest are zero); bins beyond the bitset and non-finite distances (k < 0) are kept. */
__device__ __forceinline__ bool bin_in_reach(const uint32_t *reach, uint32_t reach_words, int k)
{
    if ((uint32_t)k >= OSLAMK_REACH_BINS) return true;
    const uint32_t w = (uint32_t)k >> 5;
    return w < reach_words && ((reach[w] >> ((uint32_t)k & 31u)) & 1u);
}

/* Sizes the hit lists by demand: keep_count[ref] = pairs of the reference point whose distance bin is within
 * reach (what k_scene_hits keys and looks up), an upper bound of its hits -- 16 % above them on the bench
 * scene.  A workgroup tests its tile of scene points against COUNT_REFS reference points (their coordinates
 * sit in scalar registers), so a point is loaded once per 8 pairs.
 * grid (ceil(n_launch / COUNT_REFS), ceil(S / KEY_TILE)). */
__global__ __launch_bounds__(256) void k_scene_count(oslamk_vote_args a)
{
    __shared__ uint32_t s_cnt[COUNT_REFS];
    __shared__ uint32_t s_reach[OSLAMK_REACH_BINS / 32];
    const int S = a.scene.n, lane = threadIdx.x & (WAVE - 1);
    const int g0 = blockIdx.x * COUNT_REFS;
    const uint32_t reach_words = a.table.reach_words;
    uint32_t rr[COUNT_REFS], cnt[COUNT_REFS];
    float prx[COUNT_REFS], pry[COUNT_REFS], prz[COUNT_REFS];
#pragma unroll
    for (int g = 0; g < COUNT_REFS; g++) {
        const bool v = g0 + g < a.n_launch;
        rr[g] = v ? a.ref_idx[a.first_ref + g0 + g] : 0xffffffffu;
        prx[g] = v ? a.scene.px[rr[g]] : 0.0f;
        pry[g] = v ? a.scene.py[rr[g]] : 0.0f;
        prz[g] = v ? a.scene.pz[rr[g]] : 0.0f;
        cnt[g] = 0;
    }
    if (threadIdx.x < COUNT_REFS) s_cnt[threadIdx.x] = 0;
    for (uint32_t w = threadIdx.x; w < reach_words; w += 256) s_reach[w] = a.table.reach[w];
    __syncthreads();
    for (int c = 0; c < KEY_TILE / 256; c++) {
        const int i = blockIdx.y * KEY_TILE + c * 256 + threadIdx.x;
        const bool in = i < S;
        const float x = in ? a.scene.px[i] : 0.0f, y = in ? a.scene.py[i] : 0.0f, z = in ? a.scene.pz[i] : 0.0f;
#pragma unroll
        for (int g = 0; g < COUNT_REFS; g++) {
            const int k = pair_dist_bin_quick(x - prx[g], y - pry[g], z - prz[g], a.d_dist, a.inv_d_dist);
            const bool keep = in && (uint32_t)i != rr[g] && bin_in_reach(s_reach, reach_words, k);
            cnt[g] += (uint32_t)__popcll(__ballot(keep));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < COUNT_REFS; g++)
            if (cnt[g]) atomicAdd(&s_cnt[g], cnt[g]);
    }
    __syncthreads();
    if (threadIdx.x < COUNT_REFS && g0 + (int)threadIdx.x < a.n_launch && s_cnt[threadIdx.x])
        atomicAdd(&a.keep_count[g0 + threadIdx.x], s_cnt[threadIdx.x]);
}

/* Phase 2 of k_scene_hits for one wave: up to 64 pairs (reference point ref_local of the batch, point
 * tile0 + list[j0 + lane]) that are within reach.  The pair's quantised feature as bins (pc_pair_bins: the
 * reference's operations up to each acosf argument, then the tabulated steps), the slot of its key in the
 * union table from table.kmap -- one load instead of the hash and a probe sequence -- and for a pair that
 * hits, theta_v; the hits are appended to the reference point's list, one atomic per wave for the places. */
__device__ __forceinline__ void hits_chunk(const oslamk_vote_args &a, int ref_local, int tile0, const uint16_t *list,
                                           uint32_t j0, uint32_t n, const uint32_t *lut, int lane)
{
    const int ref_ord = a.first_ref + ref_local;
    const uint32_t r = a.ref_idx[ref_ord];
    const float prx = a.scene.px[r], pry = a.scene.py[r], prz = a.scene.pz[r];
    const float nrx = a.scene.nx[r], nry = a.scene.ny[r], nrz = a.scene.nz[r];
    const float nrn = pc_norm3(nrx, nry, nrz);
    const float *rows = a.tsg + 8 * (size_t)ref_ord;
    const uint32_t j = j0 + (uint32_t)lane;
    bool hit = false;
    uint32_t slot = 0;
    oslamk_pay pay;
    pay.theta_t22 = 0;
    pay.idx = 0;
    if (j < n) {
        const int i = tile0 + (int)list[j];
        const float x = a.scene.px[i], y = a.scene.py[i], z = a.scene.pz[i];
        const float nx = a.scene.nx[i], ny = a.scene.ny[i], nz = a.scene.nz[i];
        const float nn = pc_norm3(nx, ny, nz);
        uint32_t combo;
        const int k1 = pc_pair_bins(prx, pry, prz, nrx, nry, nrz, nrn, x, y, z, nx, ny, nz, nn, a.d_dist, a.inv_d_dist, lut, &combo);
        if ((uint32_t)k1 < a.table.kmap_bins) {
            slot = a.table.kmap[(size_t)k1 * PC_ANGLE_COMBOS + combo];
            hit = slot != OSLAMK_KMAP_NONE;
        } else {
            /* a bin the map does not cover (or pc_pair_key's generic path): hash and probe */
            const uint32_t key = pc_pair_key(prx, pry, prz, nrx, nry, nrz, nrn, x, y, z, nx, ny, nz, nn, a.d_dist, a.inv_d_dist);
            if (key != 0) {                                       /* kernel.cu:491,520 */
                const uint32_t mask = a.table.ucap - 1;
                slot = slot_of(key, a.table.ushift);
                for (uint32_t probe = 0; probe <= mask; probe++) {
                    const uint32_t k = a.table.ukeys[slot];
                    if (k == key) { hit = true; break; }
                    if (k == 0) break;
                    slot = (slot + 1) & mask;
                }
                if (hit) slot = a.table.uids[slot];                /* the key's number, as the key map gives it */
            }
        }
        if (hit) {
            const float vy = pc_row_dot(rows, x, y, z);     /* kernel.cu:334-336 */
            const float vz = pc_row_dot(rows + 4, x, y, z);
            pay.theta_t22 = pc_angle_t22(vy, vz);
            pay.idx = (uint32_t)i;
        }
    }
    const unsigned long long hm = __ballot(hit);
    if (hm) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.hit_count[ref_local], (uint32_t)__popcll(hm));
        base = readlane_u(base, 0);
        /* The list has as many places as k_scene_count found pairs within reach, by the same predicate as phase 1
         * here -- checked, not assumed: a hit that would not fit is dropped and flagged, and the host turns the flag
         * into an error instead of using lists that ran into their neighbours. */
        const uint32_t cap = a.hit_off[ref_local + 1] - a.hit_off[ref_local];
        const uint32_t at = base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (hit && at < cap) {
            const size_t pos = (size_t)a.hit_off[ref_local] + at;
            a.hit_key[pos] = slot;
            a.hit_pay[pos] = pay;
        }
        if (lane == 0 && base + (uint32_t)__popcll(hm) > cap) atomicOr(&a.counters->list_overflow, 1u);
    }
}

/* The pairs (reference point, point) of COUNT_REFS reference points and one tile of HIT_TILE scene points.
 * Phase 1 (cheap, every pair; a point is loaded once for the 8 reference points): distance bin only; pairs
 * out of reach are dropped, the rest are compacted into one LDS list per reference point.  Phase 2 (dense
 * lanes): the waves share out the lists in chunks of 64 pairs (hits_chunk); a pair that hits is appended
 * to its reference point's hit list as {slot of the key in the union table} + {theta_v, i}.  The list of
 * a reference point has keep_count[] places (k_scene_count: the same predicate), which phase 1 cannot
 * exceed.  grid (ceil(n_launch / COUNT_REFS), ceil(S / HIT_TILE)). */
__global__ __launch_bounds__(256) void k_scene_hits(oslamk_vote_args a)
{
    __shared__ uint16_t s_list[COUNT_REFS][HIT_TILE];
    __shared__ uint32_t s_n[COUNT_REFS];
    __shared__ uint32_t s_reach[OSLAMK_REACH_BINS / 32];
    __shared__ uint32_t s_lut[2 * PC_ACOS_CELLS];
    const int S = a.scene.n, lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = threadIdx.x >> 6;
    const int g0 = blockIdx.x * COUNT_REFS, tile0 = blockIdx.y * HIT_TILE;
    const uint32_t reach_words = a.table.reach_words;
    uint32_t rr[COUNT_REFS];
    float prx[COUNT_REFS], pry[COUNT_REFS], prz[COUNT_REFS];
#pragma unroll
    for (int g = 0; g < COUNT_REFS; g++) {
        const bool v = g0 + g < a.n_launch;
        rr[g] = v ? a.ref_idx[a.first_ref + g0 + g] : 0xffffffffu;
        prx[g] = v ? a.scene.px[rr[g]] : 0.0f;
        pry[g] = v ? a.scene.py[rr[g]] : 0.0f;
        prz[g] = v ? a.scene.pz[rr[g]] : 0.0f;
    }
    if (threadIdx.x < COUNT_REFS) s_n[threadIdx.x] = 0;
    for (uint32_t w = threadIdx.x; w < reach_words; w += 256) s_reach[w] = a.table.reach[w];
    s_lut[threadIdx.x] = d_acos_lut[threadIdx.x];
    __syncthreads();

    for (int c = 0; c < HIT_TILE / 256; c++) {
        const int li = c * 256 + (int)threadIdx.x, i = tile0 + li;
        const bool in = i < S;
        const float x = in ? a.scene.px[i] : 0.0f, y = in ? a.scene.py[i] : 0.0f, z = in ? a.scene.pz[i] : 0.0f;
#pragma unroll
        for (int g = 0; g < COUNT_REFS; g++) {
            const int k = pair_dist_bin_quick(x - prx[g], y - pry[g], z - prz[g], a.d_dist, a.inv_d_dist);
            const bool keep = in && (uint32_t)i != rr[g] && g0 + g < a.n_launch && bin_in_reach(s_reach, reach_words, k);
            const unsigned long long km = __ballot(keep);
            if (km) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_n[g], (uint32_t)__popcll(km));
                base = readlane_u(base, 0);
                if (keep) s_list[g][base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = (uint16_t)li;
            }
        }
    }
    __syncthreads();

    uint32_t cc = 0;
    for (int g = 0; g < COUNT_REFS && g0 + g < a.n_launch; g++) {
        const uint32_t n_g = uni_u32(s_n[g]);
        for (uint32_t j0 = 0; j0 < n_g; j0 += WAVE, cc++)
            if ((cc & 3u) == wave) hits_chunk(a, g0 + g, tile0, &s_list[g][0], j0, n_g, s_lut, lane);
    }
}

/* the hit sort (k_sort_hits) lives in oslam_sort.hip */

/* --------------------------------------------------------------------------
 * One workgroup = one (scene reference point, model slice).
 * LDS: acc[1024][32] u32 = 128 KiB (one workgroup per CU, 16 waves).
 * ComputeUniqueVotes (model.cu:95-171) without the vote list: K3/K4
 * (kernel.cu:480-554) accumulate straight into acc, and the sort/histogram/
 * threshold of model.cu:148-170 becomes the scan at the end.
 *
 * An item of work is one run of the reference point (hits that share a key) with the key's bucket
 * in this slice, found by one load of table.uinfo[slice][slot]: there is no probing here and no
 * list of items.  Its cost is chunks x hits vote iterations (a chunk = 256 entries, an iteration =
 * one hit voting with the chunk a wave holds in registers: four LDS atomics per lane).
 *   - Pre-scan, all threads: items above VOTE_GIANT iterations ("giants": the buckets of planar
 *     surfaces hold 10^4 entries) are queued in LDS.
 *   - Giants first: cut into units of one chunk x at most 16-32 hits, dealt round-robin to the
 *     16 waves.
 *   - Then the rest, dynamically: a wave takes VOTE_BLOCK runs at a time from a counter in LDS and
 *     skips the giants; the run records of the block two ahead and the bucket records of the next
 *     one are in flight.
 * No barrier between the two; the waves meet at the end.
 * The entry stream (200 GB per 5k x 100k registration, from the Infinity Cache and HBM) is what
 * has to be kept busy: with one step's loads in flight per wave the kernel sat at the latency of a
 * load per step (59 of 89 ms with the votes compiled out).  So a wave works in groups of VOTE_GROUP
 * steps: their descriptors first, then all their loads back to back (16 B per lane each), then the
 * votes, one copy of the vote loop per member so that each waits for its own registers only.
 * Workgroups b and b + 8 share an XCD (speed only): the slices of one reference point are placed
 * on one XCD, so its hit and run lists reach one L2 once.
 * ------------------------------------------------------------------------*/
#ifndef VOTE_DIAG_CORR
#define VOTE_DIAG_CORR 0
#endif
#define VOTE_QCAP 1024
#ifndef VOTE_BLOCK
#define VOTE_BLOCK 16
#endif
#ifndef VOTE_GIANT
#define VOTE_GIANT 32
#endif
#ifndef VOTE_GROUP
#define VOTE_GROUP 4           /* must divide 64 */
#endif
#define RUN_SLOT_MASK ((1u << OSLAMK_RUN_SHIFT) - 1u)

template <int MODE, int PASS>
__device__ __forceinline__ void vote_body(const oslamk_vote_args &a, const uint32_t wg)
{
    __shared__ __attribute__((aligned(16))) uint32_t acc[ACC_CELLS + ACC_TRASH_WORDS];
    __shared__ uint32_t s_wave[VOTE_THREADS / WAVE];
    __shared__ uint32_t s_wave2[VOTE_THREADS / WAVE];
    __shared__ unsigned long long s_tot[2];            /* votes; entries streamed | items << 40 */
    __shared__ unsigned long long s_sum[VOTE_THREADS / WAVE];
    __shared__ uint32_t s_redo;
    __shared__ uint32_t s_g, s_lmax, s_base, s_qn, s_next;
    __shared__ uint32_t s_tbl[32];
    __shared__ uint32_t s_q[VOTE_QCAP];
    __shared__ SlowCtx s_ctx;
    __shared__ unsigned long long s_slow[MODE == 0 ? (VOTE_THREADS / WAVE) * SLOW_CAP : 1];

    typedef VoteRegs<MODE> VR;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const uint32_t nsl = (uint32_t)a.table.n_slices;
    const uint32_t xg = wg & 7u, xi = wg >> 3;
    const int ref_local = (int)((xi / nsl) * 8u + xg);
    const int slice = (int)(xi % nsl);
    if (ref_local >= a.n_launch) return;             /* the grid is padded to a multiple of 8 reference points */
    const int ref_ord = a.first_ref + ref_local;
    const uint32_t r = a.ref_idx[ref_ord];
    const uint32_t n_runs = a.run_count[ref_local];
    const size_t off = a.hit_off[ref_local];
    const oslamk_pay *hits = a.hit_sorted + off;
    const oslamk_run *runs = a.runs + off;
    const oslamk_uinfo *uinfo = a.table.uinfo + (size_t)slice * a.table.uinfo_stride;
    const uint32_t *e4 = a.ent.e4;
    const uint32_t m_base = (uint32_t)slice * OSLAMK_SLICE;   /* first model reference of the slice */
    if (tid == 0) {
        s_ctx.pw = a.ent.pw;
        s_ctx.puv = a.ent.puv;
        s_ctx.hits = hits;
        s_ctx.px = a.scene.px;
        s_ctx.py = a.scene.py;
        s_ctx.pz = a.scene.pz;
        s_ctx.rows = a.tsg + 8 * (size_t)ref_ord;
        s_ctx.inc_lo = PASS == 2 ? 0u : 1u;
        s_ctx.inc_hi = PASS == 0 ? 0x10000u : PASS == 2 ? 1u : 0u;
        s_ctx.dropped = 0;
    }
    const SlowCtx *sc = &s_ctx;

    for (int c = tid; c < ACC_CELLS / 4; c += VOTE_THREADS) reinterpret_cast<uint4 *>(acc)[c] = make_uint4(0, 0, 0, 0);
    if (tid < 32) s_tbl[tid] = k_alpha_thr[tid];

    unsigned long long *sq = s_slow + (MODE == 0 ? wid * SLOW_CAP : 0);   /* this wave's re-evaluation queue */
    uint32_t sq_n = 0;
#ifdef VOTE_PROF
    const long long pt0 = clock64();
#endif

    /* ---- pre-scan: votes of this workgroup, giants into the queue ---- */
    uint32_t T = VOTE_GIANT;
    for (bool first = true;; first = false) {
        if (tid == 0) {
            s_qn = 0;
            s_next = 0;
            if (first) s_tot[0] = s_tot[1] = 0;
        }
        __syncthreads();
        for (uint32_t k0 = 0; k0 < n_runs; k0 += VOTE_THREADS) {
            const uint32_t k = k0 + (uint32_t)tid;
            bool giant = false;
            unsigned long long lv = 0, le = 0;
            if (k < n_runs) {
                const oslamk_run rr = runs[k];
                const uint32_t lf = uinfo[rr.slot_r & RUN_SLOT_MASK].len;
                const uint32_t ln = lf & 0x7fffffffu, R = (rr.slot_r >> OSLAMK_RUN_SHIFT) + 1u;
                lv = (unsigned long long)ln * R;
                le = ((unsigned long long)(ln != 0u) << 40) + ln;       /* items in the upper bits: one reduction for both */
                /* items with a marker (MODE 0) are neither giants nor small items: they wait for the pass after the loop */
                giant = ((ln + 255u) >> 8) * R > T && !(MODE == 0 && ((lf | rr.first) >> 31));
            }
            if (first) {            /* summed over the wave and parked in LDS: no register lives on through the vote loops */
                lv = wave_sum_u64(lv);
                le = wave_sum_u64(le);
                if (lane == 0 && le) {
                    atomicAdd(&s_tot[0], lv);
                    atomicAdd(&s_tot[1], le);
                }
            }
            const unsigned long long gm = __ballot(giant);
            if (gm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_qn, (uint32_t)__popcll(gm));
                base = readlane_u(base, 0);
                const uint32_t p = base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull));
                if (giant && p < VOTE_QCAP) s_q[p] = k;
            }
        }
        __syncthreads();            /* also: acc zeroed, s_tbl and s_ctx written */
        if (s_qn <= VOTE_QCAP) break;
        T *= 2;                     /* more giants than the queue holds: a higher bar, same for every thread */
        __syncthreads();
    }
    const uint32_t n_q = uni_u32(s_qn);
    T = uni_u32(T);
#ifdef VOTE_PROF
    const long long ptA = clock64();
#endif

    /* A counter word of the accumulator holds two 16-bit counters (two model reference points per row: a slice of
     * 2046 fits the 128 KiB).  PASS 0 -- the kernel every registration runs -- votes into both at once.  A counter
     * that passes 65535 carries into its neighbour: the sum of all counters then falls short of the votes cast
     * (every overflow takes 65535 or 65536 away, never adds); the workgroup notices, emits nothing and puts itself
     * on the redo list, and k_vote_wide votes for it again with full 32-bit counters, first for the lower half of
     * the slice's model points (PASS 1), then for the upper (PASS 2).  Counts beyond 16 bits need large planar
     * surfaces in both clouds; the bench scene's largest cell is 17 186. */
    const uint32_t pass = PASS;
    const uint32_t inc_lo = PASS == 2 ? 0u : 1u, inc_hi = PASS == 0 ? 0x10000u : PASS == 2 ? 1u : 0u;
    {
        /* ---- the work of this wave, as windows of up to 64 step descriptors made lane-parallel ----
         * A set of items sits in the lanes of two register pairs: the run record {slot | hits - 1, first hit} and
         * the bucket record {start, length} of up to 64 items; `take` says which lanes belong to the set, and
         * every item is cut into units (chunks, times 1, 2 or 4 ranges of hits for a giant).  The wave takes the
         * units u0, u0 + stride, ... of the set: lane i of a window looks its unit up in the running sums of the
         * set (binary search across lanes with ds_bpermute) and computes the step {first entry, entries left,
         * first hit, hits, hit range} -- about one instruction per step instead of a scalar generator of ~60.
         * The steps of a window are then voted in groups of VOTE_GROUP: descriptors to scalar registers,
         * all the group's loads back to back (16 B per lane each), the votes, one copy of the vote loop
         * per member so that each waits for its own registers only. */
        uint2 B_ru = make_uint2(0, 0), B_inf = make_uint2(0, 0);
        auto bperm = [&](uint32_t from_lane, uint32_t v) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)v);
        };
        /* step g of a window: its descriptor from lane g of the window's registers (VOTE_GROUP divides 64, so a
         * group never runs past lane 63; lanes past the window's end describe empty steps) */
        auto step_at = [&](uint32_t ve0, uint32_t vh0, uint32_t vmisc, int l) -> VoteStep {
            VoteStep d;
            const uint32_t misc = readlane_u(vmisc, l);
            d.e0 = readlane_u(ve0, l);
            d.h0 = readlane_u(vh0, l);
            d.R = (misc & 63u) + 1u;
            d.i0 = (misc >> 6) & 127u;
            d.i1 = (misc >> 13) & 127u;
            d.left = misc >> 20;            /* capped at a chunk: all a step asks is which lanes hold entries */
            d.valid = true;
            return d;
        };
        /* hit ranges a chunk of an item with R hits is cut into, as a shift: giants only */
        auto split_of = [&](uint32_t R, bool split) -> uint32_t { return !split ? 0u : R > 32u ? 2u : R > 16u ? 1u : 0u; };
        /* votes the units u0, u0 + stride, ... of the items in the lanes where `take` holds */
        auto vote_set = [&](bool take, bool split, uint32_t u0, uint32_t stride) {
            uint32_t excl, incl;
            {
                const uint32_t R = (B_ru.x >> OSLAMK_RUN_SHIFT) + 1u;
                const uint32_t nu = take ? (((B_inf.y & 0x7fffffffu) + 255u) >> 8) << split_of(R, split) : 0u;
                incl = nu;
                for (int o = 1; o < WAVE; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o, WAVE);
                    if (lane >= o) incl += up;
                }
                excl = incl - nu;
            }
            const uint32_t total = readlane_u(incl, WAVE - 1);
            for (uint32_t ub = u0; ub < total; ub += stride * WAVE) {
                /* ---- the window: lane i <-> unit ub + stride * i ---- */
                const uint32_t u = ub + stride * (uint32_t)lane;
                const bool act = u < total;
                uint32_t lo = 0;                                   /* items whose running sum is <= u: the unit's item */
    #pragma unroll
                for (uint32_t s2 = WAVE / 2; s2 > 0; s2 >>= 1) lo += bperm(lo + s2 - 1u, incl) <= u ? s2 : 0u;
                lo = lo < (uint32_t)WAVE - 1u ? lo : (uint32_t)WAVE - 1u;
                const uint32_t st = bperm(lo, B_inf.x), ln = bperm(lo, B_inf.y) & 0x7fffffffu, h0 = bperm(lo, B_ru.y) & 0x7fffffffu;
                const uint32_t R = (bperm(lo, B_ru.x) >> OSLAMK_RUN_SHIFT) + 1u, lg = split_of(R, split);
                const uint32_t local = u - bperm(lo, excl);
                const uint32_t c = local >> lg, hs = local & ((1u << lg) - 1u), per = (R + (1u << lg) - 1u) >> lg;
                const uint32_t i0 = hs * per, i1 = i0 + per < R ? i0 + per : R;
                /* a lane past the end describes a step that loads entry 0 and hit 0 and votes nothing */
                const uint32_t left = ln - (c << 8);
                const uint32_t w_e0 = act ? st + (c << 8) : 0u, w_h0 = act ? h0 : 0u,
                               w_misc = act ? (R - 1u) | (i0 << 6) | (i1 << 13) | ((left < 256u ? left : 256u) << 20) : 0u;
                const uint32_t rest = (total - ub + stride - 1u) / stride, n_w = rest < (uint32_t)WAVE ? rest : (uint32_t)WAVE;
                for (uint32_t g0 = 0; g0 < n_w; g0 += VOTE_GROUP) {
                    VR regs[VOTE_GROUP];
                    VoteStep ds[VOTE_GROUP];
    #pragma unroll
                    for (int g = 0; g < VOTE_GROUP; g++) {
                        ds[g] = step_at(w_e0, w_h0, w_misc, (int)g0 + g);
                        regs[g].load(e4, hits, ds[g], lane);
                    }
                    asm volatile("" ::: "memory");      /* all the group's loads are issued here, in this order: none sinks into its vote */
    #pragma unroll
                    for (int g = 0; g < VOTE_GROUP; g++) regs[g].vote(acc, ds[g], lane, inc_lo, inc_hi);
                }
            }
        };
        /* Exact mode, after the votes of a set of items: the votes that lie within the margin of a bin edge, found
         * instead of tested for.  A vote of hit h with entry word w has position (P(base_h) - P(w)) mod 2^32 inside
         * its (shifted) bin, P(x) = x * 30 mod 2^32 -- the low word of vote_product -- and is near an edge when that
         * is below PC_T24_EDGE.  The bucket exists a second time ordered by P(w) (oslamk_entries.pw, segments of
         * OSLAMK_PSEG entries), so the near-edge entries of a hit are a contiguous piece of it, cyclically: from the
         * first entry with P(w) >= A = P(base_h) - PC_T24_EDGE + 1 on, as long as (P(w) - A) mod 2^32 < PC_T24_EDGE.
         * One lane per (hit, bucket segment) pair of the set: a binary search, then the candidates -- one vote in
         * 4000 -- go to the wave's queue, whose flush moves those that the reference's float sequence bins otherwise.
         * u0 / stride: which of the set's pairs this wave takes (giants are shared out between the waves). */
        auto correct_set = [&](bool take, uint32_t u0, uint32_t stride) {
            const uint32_t *pw = a.ent.pw;
            const uint16_t *pdir = a.ent.pdir;
            uint32_t excl, incl;
            {
                const uint32_t R = (B_ru.x >> OSLAMK_RUN_SHIFT) + 1u, ln = B_inf.y & 0x7fffffffu;
                const uint32_t np = take ? R * ((ln + OSLAMK_PSEG - 1u) / OSLAMK_PSEG) : 0u;
                incl = np;
                for (int o = 1; o < WAVE; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o, WAVE);
                    if (lane >= o) incl += up;
                }
                excl = incl - np;
            }
            const uint32_t total = readlane_u(incl, WAVE - 1);
            for (uint32_t pb = u0; pb < total; pb += stride * WAVE) {
                const uint32_t p = pb + stride * (uint32_t)lane;
                const bool act = p < total;
                uint32_t lo = 0;                                   /* items whose running sum is <= p: the pair's item */
    #pragma unroll
                for (uint32_t s2 = WAVE / 2; s2 > 0; s2 >>= 1) lo += bperm(lo + s2 - 1u, incl) <= p ? s2 : 0u;
                lo = lo < (uint32_t)WAVE - 1u ? lo : (uint32_t)WAVE - 1u;
                const uint32_t st = bperm(lo, B_inf.x), ln = bperm(lo, B_inf.y) & 0x7fffffffu, h0 = bperm(lo, B_ru.y) & 0x7fffffffu;
                const uint32_t R = (bperm(lo, B_ru.x) >> OSLAMK_RUN_SHIFT) + 1u;
                uint32_t local = p - bperm(lo, excl), seg = 0;
                if (__ballot(act && local >= R)) {               /* buckets above OSLAMK_PSEG entries: rare */
                    seg = act ? local / R : 0u;
                    local -= seg * R;
                }
                const uint32_t hit = act ? h0 + local : 0u;
                const uint32_t sbase = act ? st + seg * OSLAMK_PSEG : 0u;
                const uint32_t n = act ? (ln - seg * OSLAMK_PSEG < OSLAMK_PSEG ? ln - seg * OSLAMK_PSEG : OSLAMK_PSEG) : 0u;
#if VOTE_DIAG_CORR == 1                /* timing-only build: the enumeration of the pairs alone */
                asm volatile("" ::"v"(hit), "v"(sbase), "v"(n));
                continue;
#endif
                const uint32_t A = pc_vote_base_t32(hits[hit].theta_t22) * 30u - (PC_T24_EDGE - 1u);
                /* the cells the window [A, A + PC_T24_EDGE) touches, through the segment's directory: positions
                 * d0 .. d0 + cnt - 1 (cyclic) hold every near-edge entry, and a few that are not */
                const uint32_t K = OSLAMK_PDIR_CELLS(n), B = A + (PC_T24_EDGE - 1u);
                const uint32_t cA = __umulhi(A, K), cB = __umulhi(B, K);
                const uint32_t d0 = act ? pdir[sbase + cA] : 0u, d1 = act ? pdir[sbase + cB + 1u] : 0u;
                const uint32_t cnt = K == 1u ? n : B < A ? n - d0 + d1 : d1 - d0;       /* B < A: the window wraps past 2^32 */
#if VOTE_DIAG_CORR == 2                /* timing-only build: up to the directory, without the entries */
                asm volatile("" ::"v"(cnt), "v"(d0));
                continue;
#endif
                for (uint32_t c0 = 0; __ballot(c0 < cnt); c0 += 4u) {
                    uint32_t k[4], w[4];
    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        k[j] = d0 + c0 + (uint32_t)j;
                        k[j] = k[j] >= n ? k[j] - n : k[j];
                        w[j] = pw[sbase + (c0 + (uint32_t)j < cnt ? k[j] : 0u)];
                    }
    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned long long nm = __ballot(c0 + (uint32_t)j < cnt && w[j] * 30u - A < PC_T24_EDGE);
                        if (nm) sq_n = slow_push((lds_ctx *)sc, (lds_u32 *)acc, (lds_u32 *)s_tbl, (lds_u64 *)sq, sq_n, nm, (sbase + k[j]) | SLOW_VERIFY, hit, lane);
                    }
                }
            }
        };
        auto marked = [&](const uint2 &ru, const uint2 &inf) -> bool { return MODE == 0 && ((inf.y | ru.y) >> 31); };

        /* ---- the sets, one after the other (one call site of vote_set: its vote loops exist once) ----
         * giants first: 64 queued items at a time, their units dealt round-robin to the waves;
         * then the rest, dynamically: VOTE_BLOCK runs at a time from a counter in LDS.  Four blocks sit side by
         * side in the lanes of B_ru / B_inf (block b in lanes 16 * (b's turn mod 4) ..): the one being voted, the
         * next (its bucket records in flight) and the one after it (its run records in flight) */
        {
            const uint32_t n_blocks = (n_runs + VOTE_BLOCK - 1) / VOTE_BLOCK;
            const uint32_t my_slot = (uint32_t)lane / VOTE_BLOCK, k_in = (uint32_t)lane % VOTE_BLOCK;
            auto grab = [&]() -> uint32_t {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&s_next, 1u);
                return uni_u32(readlane_u(b, 0));
            };
            /* run records of block b into the lanes of `slot` (zeros past the end of the run list) */
            auto fetch_runs = [&](uint32_t b, uint32_t slot) {
                if (my_slot == slot) {
                    const uint32_t k = b * VOTE_BLOCK + k_in;
                    B_ru = make_uint2(0, 0);
                    B_inf = make_uint2(0, 0);
                    if (k < n_runs) {
                        const oslamk_run rr = runs[k];
                        B_ru = make_uint2(rr.slot_r, rr.first);
                    }
                }
            };
            /* bucket records of the block in `slot` (its run records have landed) */
            auto fetch_info = [&](uint32_t b, uint32_t slot) {
                if (my_slot == slot && b * VOTE_BLOCK + k_in < n_runs) {
                    const oslamk_uinfo ui = uinfo[B_ru.x & RUN_SLOT_MASK];
                    B_inf = make_uint2(ui.start, ui.len);
                }
            };
            uint32_t cur_b = 0, nx1_b = 0, nx2_b = 0, turn = 0;
            uint32_t tw = 0;                                 /* giants: first queue entry of the next set */
            bool started = false;
            for (;;) {
                bool take, split;
                uint32_t u0, stride;
                if (tw < n_q) {
                    const uint32_t t = tw + (uint32_t)lane;
                    B_ru = make_uint2(0, 0);
                    B_inf = make_uint2(0, 0);
                    if (t < n_q) {
                        const oslamk_run rr = runs[s_q[t]];
                        const oslamk_uinfo ui = uinfo[rr.slot_r & RUN_SLOT_MASK];
                        B_ru = make_uint2(rr.slot_r, rr.first);
                        B_inf = make_uint2(ui.start, ui.len);
                    }
                    tw += WAVE;
                    take = t < n_q;
                    split = true;
                    u0 = uni_u32((uint32_t)wid);
                    stride = VOTE_THREADS / WAVE;
                } else {
                    if (!started) {                          /* the first three blocks of this wave */
                        started = true;
                        cur_b = grab();
                        fetch_runs(cur_b, 0u);
                        nx1_b = grab();
                        fetch_runs(nx1_b, 1u);
                        nx2_b = grab();
                        fetch_runs(nx2_b, 2u);
                        fetch_info(cur_b, 0u);
                        fetch_info(nx1_b, 1u);
                    }
                    if (cur_b >= n_blocks) break;            /* the counter only grows: this wave is done */
                    const uint32_t ln = B_inf.y & 0x7fffffffu, R = (B_ru.x >> OSLAMK_RUN_SHIFT) + 1u;
                    /* of the current block: present in the slice, not a giant (those are done), no marker (those come last) */
                    take = my_slot == (turn & 3u) && ln != 0u && ((ln + 255u) >> 8) * R <= T && !marked(B_ru, B_inf);
                    split = false;
                    u0 = 0u;
                    stride = 1u;
                    /* the next blocks move up before the votes, so that their loads fly meanwhile */
                    cur_b = nx1_b;
                    nx1_b = nx2_b;
                    fetch_info(nx1_b, (turn + 2u) & 3u);
                    nx2_b = grab();
                    fetch_runs(nx2_b, (turn + 3u) & 3u);
                    turn++;
                }
                vote_set(take, split, u0, stride);
#ifndef VOTE_DIAG_NOCORRECT          /* timing-only build: exact mode without the near-edge search */
                if (MODE == 0) correct_set(take, u0, stride);
#endif
            }
        }
        if (MODE == 0) {
            /* the items with a marker, dealt round-robin to the waves: every vote through the queue */
            for (uint32_t k0 = (uint32_t)wid * WAVE; k0 < n_runs; k0 += VOTE_THREADS) {
                const uint32_t k = k0 + (uint32_t)lane;
                oslamk_run rr;
                oslamk_uinfo ui;
                rr.slot_r = rr.first = 0;
                ui.start = ui.len = 0;
                if (k < n_runs) {
                    rr = runs[k];
                    ui = uinfo[rr.slot_r & RUN_SLOT_MASK];
                }
                unsigned long long fm = __ballot((ui.len & 0x7fffffffu) != 0u && ((ui.len | rr.first) >> 31));
                while (fm) {
                    const int j = __ffsll((long long)fm) - 1;
                    fm &= fm - 1ull;
                    sq_n = forced_item((lds_ctx *)sc, (lds_u32 *)acc, (lds_u32 *)s_tbl, (lds_u64 *)sq, sq_n, readlane_u(ui.start, j),
                                       readlane_u(ui.len, j) & 0x7fffffffu, readlane_u(rr.first, j) & 0x7fffffffu,
                                       (readlane_u(rr.slot_r, j) >> OSLAMK_RUN_SHIFT) + 1u, lane);
                }
            }
            if (sq_n) slow_queue_flush((lds_ctx *)sc, (lds_u32 *)acc, (lds_u32 *)s_tbl, (lds_u64 *)sq, sq_n, lane);
        }
#ifdef VOTE_PROF
        const long long pt1 = clock64();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the atomics issued from asm (VoteRegs::vote) */
        __syncthreads();
#ifdef VOTE_PROF
        const long long pt2 = clock64();
        if (lane == 0) {             /* wave cycles: pre-scan (with zeroing), voting, wait at the barrier behind it */
            atomicAdd(&a.counters->prof[0], (unsigned long long)(ptA - pt0));
            atomicAdd(&a.counters->prof[1], (unsigned long long)(pt1 - ptA));
            atomicAdd(&a.counters->prof[2], (unsigned long long)(pt2 - pt1));
        }
#endif


        /* ---- peak extraction: local max, non-empty cells, sum, emission ---- */
        /* the two counters of word w: of the model points m_base + row and m_base + 1023 + row */
        auto lower_of = [&](uint32_t w) -> uint32_t { return pass == 0 ? w & 0xffffu : pass == 1 ? w : 0u; };
        auto upper_of = [&](uint32_t w) -> uint32_t { return pass == 0 ? w >> 16 : pass == 2 ? w : 0u; };
        uint32_t lmax = 0, nz = 0;
        unsigned long long sum = 0;
#pragma unroll 2
        for (int c = tid; c < ACC_REAL_CELLS; c += VOTE_THREADS) {
            const uint32_t w = acc[c], v0 = lower_of(w), v1 = upper_of(w);
            lmax = max(lmax, max(v0, v1));
            nz += (v0 != 0) + (v1 != 0);
            sum += (unsigned long long)v0 + v1;
        }
        lmax = wave_max_u32(lmax);
        nz = wave_sum_u32(nz);
        sum = wave_sum_u64(sum);
        if (lane == 0) {
            s_wave[wid] = lmax;
            s_wave2[wid] = nz;
            s_sum[wid] = sum;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t m = 0, n = 0;
            unsigned long long tot = 0;
            for (int w = 0; w < VOTE_THREADS / WAVE; w++) {
                m = s_wave[w] > m ? s_wave[w] : m;
                n += s_wave2[w];
                tot += s_sum[w];
            }
            /* every vote cast is in a counter unless a 16-bit counter overflowed */
            s_redo = pass == 0 && tot != s_tot[0] - s_ctx.dropped;
#if defined(VOTE_DIAG_NOLOOP) || defined(VOTE_DIAG_NOATOM)
            s_redo = 0;                 /* timing-only builds cast no votes: nothing to check */
#endif
            if (s_redo) {
                a.redo[atomicAdd(&a.counters->redo_count, 1u)] = wg;
                atomicAdd(&a.counters->redo_total, 1u);
            }
            if (!s_redo) {
                if (pass <= 1) {                /* once per workgroup */
                    /* hits are counted once per reference point */
                    const unsigned long long h = slice == 0 ? (unsigned long long)a.hit_count[ref_local] : 0ull;
                    if (h) atomicAdd(&a.counters->hits, h);
                    if (s_tot[0]) atomicAdd(&a.counters->votes, s_tot[0]);
                    if (s_tot[1]) {
                        atomicAdd(&a.counters->entries, s_tot[1] & 0xffffffffffull);
                        atomicAdd(&a.counters->items, s_tot[1] >> 40);
                    }
                }
                uint32_t g = a.fixed_gmax;
                if (g == 0) {
                    const uint32_t old = atomicMax(&a.counters->gmax, m);
                    g = old > m ? old : m;
                }
                if (n) atomicAdd(&a.counters->nonzero_cells, (unsigned long long)n);
                s_g = g;
                s_lmax = m;
            }
        }
        __syncthreads();
        if (s_redo) return;                         /* workgroup-uniform: k_vote_wide takes over */

        if (a.acc_dump && ref_ord == a.dump_ref) {
            uint32_t *dst = a.acc_dump + (size_t)m_base * OSLAMK_NBIN;
    #pragma unroll 2
        for (int c = tid; c < ACC_REAL_CELLS; c += VOTE_THREADS) {
                const uint32_t w = acc[c];
                const int cell = c / ACC_STRIDE * OSLAMK_NBIN + c % ACC_STRIDE;
                if (pass <= 1) dst[cell] = lower_of(w);
                if (pass != 1) dst[cell + (int)PC_ROWS_PER_HALF * OSLAMK_NBIN] = upper_of(w);
                if (ACC_STRIDE < OSLAMK_NBIN && c % ACC_STRIDE == ACC_STRIDE - 1) {      /* the dump's last column */
                    if (pass <= 1) dst[cell + 1] = 0;
                    if (pass != 1) dst[cell + 1 + (int)PC_ROWS_PER_HALF * OSLAMK_NBIN] = 0;
                }
            }
        }

        /* cells with count > thresh * g (model.cu:164-167; g <= final maximum, so
         * this is a superset that the host filters with the final maximum) */
        const float bound = a.thresh * (float)s_g;
        if ((float)s_lmax > bound) {                 /* workgroup-uniform */
            uint32_t cnt = 0;
    #pragma unroll 2
        for (int c = tid; c < ACC_REAL_CELLS; c += VOTE_THREADS) {
                const uint32_t w = acc[c];
                cnt += ((float)lower_of(w) > bound) + ((float)upper_of(w) > bound);
            }
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
    #pragma unroll 2
        for (int c = tid; c < ACC_REAL_CELLS; c += VOTE_THREADS) {
                const uint32_t w = acc[c];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const uint32_t v = hf ? upper_of(w) : lower_of(w);
                    if ((float)v > bound) {
                        if (pos < a.out_cap) {
                            oslamk_cell cell;
                            const uint32_t m_r = m_base + (uint32_t)hf * PC_ROWS_PER_HALF + (uint32_t)(c / ACC_STRIDE);
                            cell.code = ((unsigned long long)r << 32) | (unsigned long long)((m_r << 6) |
                                                                         (uint32_t)(c % ACC_STRIDE));
                            cell.count = v;
                            cell.pad = 0;
                            a.out[pos] = cell;
                        }
                        pos++;
                    }
                }
            }
        }
#ifdef VOTE_PROF
        if (lane == 0) atomicAdd(&a.counters->prof[3], (unsigned long long)(clock64() - pt2));     /* peak extraction */
#endif
    }
}

template <int MODE>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote(oslamk_vote_args a)
{
    vote_body<MODE, 0>(a, blockIdx.x);
}

/* the workgroups whose 16-bit counters overflowed, again with 32-bit counters for one half of the slice's model
 * points (see vote_body); a small fixed grid walks the redo list, which is almost always empty */
#define VOTE_WIDE_GRID 256
template <int MODE, int PASS>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote_wide(oslamk_vote_args a)
{
    const uint32_t n = a.counters->redo_count;
#ifdef WIDE_NOLOOP_EXPERIMENT
    if (blockIdx.x < n) vote_body<MODE, PASS>(a, a.redo[blockIdx.x]);
#else
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        vote_body<MODE, PASS>(a, a.redo[i]);
        __syncthreads();                            /* the LDS of this workgroup is reused by the next entry */
    }
#endif
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
    /* the 26 neighbour cells in the reference's (dx, dy, dz) order = ascending c; lane c finds the start of
     * cell c's poses in the sorted list (26 binary searches side by side instead of one after the other:
     * with 10^6 poses the dependent loads of the searches were most of this kernel's time) */
    uint32_t my_h = 0;
    int my_lo = n;
    if (lane < 27 && lane != 13) {
        const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
        my_h = fnv_cell3(cx + dx, cy + dy, cz + dz);
        if (my_h != 0) {                                          /* a hash of 0 is never searched (kernel.cu:727) */
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if (shash[mid] < my_h) lo = mid + 1; else hi = mid;
            }
            my_lo = lo;
        }
    }
    for (int c = 0; c < 27; c++) {
        if (c == 13) continue;                                   /* kernel.cu:684-689 */
        const uint32_t h = readlane_u(my_h, c);
        if (h == 0) continue;
        const int lo = (int)readlane_u((uint32_t)my_lo, c);
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
            while (m) {                                          /* ascending j: the reference's order */
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                votes += readlane_f(w, b);
            }
            if (__ballot(in_cell) != ~0ull) break;               /* the cell's run ended in this step */
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

int oslamk_bucket_spread(oslamk_table t, oslamk_entries ent, void *stream)
{
    const size_t total = (size_t)t.n_slices * t.cap;
    hipLaunchKernelGGL(k_bucket_spread, dim3((unsigned)total), dim3(SPREAD_THREADS), 0, (hipStream_t)stream, t, ent);
    return (int)hipGetLastError();
}

int oslamk_bucket_psort(oslamk_table t, oslamk_entries ent, void *stream)
{
    const size_t total = (size_t)t.n_slices * t.cap;
    hipLaunchKernelGGL(k_bucket_psort, dim3((unsigned)total), dim3(SPREAD_THREADS), 0, (hipStream_t)stream, t, ent);
    return (int)hipGetLastError();
}

int oslamk_reach_build(oslamk_table t, float d_dist, void *stream)
{
    hipLaunchKernelGGL(k_reach_build, dim3(OSLAMK_REACH_BINS), dim3(256), 0, (hipStream_t)stream, t, d_dist);
    return (int)hipGetLastError();
}

int oslamk_union_ids(oslamk_table t, uint32_t *counter, void *stream)
{
    hipLaunchKernelGGL(k_union_ids, dim3((t.ucap + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, t, counter);
    return (int)hipGetLastError();
}

int oslamk_kmap_build(oslamk_table t, float d_dist, void *stream)
{
    if (t.kmap_bins == 0) return 0;
    hipLaunchKernelGGL(k_kmap_build, dim3(t.kmap_bins), dim3(256), 0, (hipStream_t)stream, t, d_dist);
    return (int)hipGetLastError();
}

int oslamk_uinfo_build(oslamk_table t, void *stream)
{
    const size_t total = (size_t)t.n_slices * t.cap;
    hipLaunchKernelGGL(k_uinfo_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

int oslamk_scene_count(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    dim3 grid((unsigned)((a->n_launch + COUNT_REFS - 1) / COUNT_REFS), (unsigned)((a->scene.n + KEY_TILE - 1) / KEY_TILE));
    hipLaunchKernelGGL(k_scene_count, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

int oslamk_scene_hits(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    dim3 grid((unsigned)((a->n_launch + COUNT_REFS - 1) / COUNT_REFS), (unsigned)((a->scene.n + HIT_TILE - 1) / HIT_TILE));
    hipLaunchKernelGGL(k_scene_hits, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}


int oslamk_vote(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    /* padded to groups of 8 reference points: see the workgroup -> (reference point, slice) map in k_vote */
    dim3 grid((unsigned)(((size_t)a->n_launch + 7) / 8 * 8 * a->table.n_slices));
    if (a->mode == 0) {
        hipLaunchKernelGGL(k_vote<0>, grid, dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<0, 1>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<0, 2>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    } else {
        hipLaunchKernelGGL(k_vote<1>, grid, dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<1, 1>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<1, 2>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    }
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
