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

#include "oslam_vote_body.inc"

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
            ent.pdir[(base << OSLAMK_PDIR_SHIFT) + k] = (uint16_t)lo;
        }
        __syncthreads();
    }
}

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

template <int MODE>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote(oslamk_vote_args a)
{
    vote_body<MODE, 0>(a, blockIdx.x);
}
/* The votes of all the members of a database group in ONE grid: member j = blockIdx / workgroups per member, its
 * arguments from an array in device memory (copied to scalar registers as a kernel argument would be).  Every member
 * votes the same batch of reference points; a member with fewer slices than the widest leaves its surplus
 * workgroups at once (vote_body's own test of the reference point).  Fifty models on a depth frame are fifty grids of
 * some 280 workgroups otherwise -- each a little more than one round on 256 CUs, each with its own tail. */
template <int MODE>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote_group(const oslamk_vote_args *all, uint32_t wgs_per_member)
{
    const uint32_t j = blockIdx.x / wgs_per_member;
    const oslamk_vote_args a = all[j];
    vote_body<MODE, 0>(a, blockIdx.x - j * wgs_per_member);
}
/* k_vote_wide (the re-vote of workgroups whose 16-bit counters overflowed) lives in oslam_vote_wide.hip */

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

/* The cells of the sorted pose list as a table: cell hash -> [first, end) of its poses in sorted order (open addressing,
 * at most half full; tab[3 * slot] = hash, 0 = empty: a hash of 0 is never searched).  Two small kernels: the first
 * pose of every cell claims a slot and writes `first`; then the first pose of the NEXT cell (or the end of the list)
 * finds that slot again and writes `end`. */
__device__ __forceinline__ uint32_t cell_slot(uint32_t h, uint32_t mask) { return (h * 2654435761u) & mask; }

__global__ void k_cell_table_first(const uint32_t *shash, int n, uint32_t *tab, uint32_t mask)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t h = shash[j];
    if (h == 0u || (j > 0 && shash[j - 1] == h)) return;
    for (uint32_t s = cell_slot(h, mask);; s = (s + 1u) & mask) {
        const uint32_t old = atomicCAS(&tab[3u * s], 0u, h);
        if (old == 0u) { tab[3u * s + 1u] = (uint32_t)j; return; }
    }
}

__global__ void k_cell_table_end(const uint32_t *shash, int n, uint32_t *tab, uint32_t mask)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;     /* j = 1 .. n: the position behind a cell's last pose */
    if (j < 1 || j > n) return;
    const uint32_t h = shash[j - 1];
    if (h == 0u || (j < n && shash[j] == h)) return;
    for (uint32_t s = cell_slot(h, mask);; s = (s + 1u) & mask) {
        const uint32_t k = tab[3u * s];
        if (k == h) { tab[3u * s + 2u] = (uint32_t)j; return; }
        if (k == 0u) return;                                  /* not reached: every cell was entered */
    }
}

/* One wave per FOUR consecutive poses of the sorted list.  Lane c < 27 looks the neighbour cell c up in the table (the
 * reference's (dx, dy, dz) order = ascending c; the pose's own cell is skipped, kernel.cu:684-689); the poses of the 26
 * cells then form one list, cell after cell, which the 64 lanes test 64 at a time, four rounds' loads in flight -- a
 * handful of independent loads per pose instead of 26 binary searches and 26 dependent cell walks (with 5 * 10^4 poses
 * per model of the depth-stream database the kernel sat at the latency of those chains: 0.5 ms per model).
 *
 * The compatible poses are added by walking the ballot mask in ascending order, i.e. exactly the sequential float sum
 * of the reference loop -- or, when the weighted votes are whole numbers with a small sum (`whole`), lane-parallel.  In
 * that case the wave's poses that lie in one cell (neighbours in the sorted list) also SHARE the walk: they have the
 * same 26 neighbour cells, so a candidate is loaded once and tested against up to four poses -- a dense cluster (an
 * instance in the scene: 10^4 poses in a few cells, every one walking all the others) is bound by the candidates'
 * bytes from L2, not by the tests.  sq/st/sw are the pose quaternions, translations and weighted votes in sorted
 * (cell hash, pose index) order, sidx the pose index of a sorted position, cell the cell coordinates in pose order. */
__global__ __launch_bounds__(64) void k_cluster_scores(int n, const int *cell, const uint32_t *shash, const uint32_t *sidx,
                                                       const uint32_t *tab, uint32_t mask,
                                                       const float4 *sq, const float *st, const float *sw,
                                                       float dist2_below, int use_l1, float *score, int whole_host,
                                                       const unsigned long long *whole_dev)
{
    const int lane = threadIdx.x;
    const uint32_t p0 = 4u * blockIdx.x, un = (uint32_t)n;
    /* Weighted votes that are all whole numbers and sum to less than 2^24 (weights of 1, the default: the votes are
     * counts) add up exactly in float whatever the order: every partial sum is a whole number below 2^24.  Then the
     * lanes keep their own sums and the wave adds them once -- the same bits as the reference's sequential sum, without
     * walking the compatible poses one by one (a dense cluster has thousands per pose).  Decided by the caller for
     * host-made input (whole_host) or by k_pose_cells on the device (whole_dev[0] = sum of the votes, [1] != 0: one
     * of them is not whole). */
    const bool whole = whole_dev ? (whole_dev[1] == 0ull && whole_dev[0] < (1ull << 24) - 1ull) : whole_host != 0;
    const float rot_thresh = 2 * PM_D_ANGLE, rot_thresh_sq = rot_thresh * rot_thresh;
    for (uint32_t g0 = 0; g0 < 4u && p0 + g0 < un;) {
        /* the group: this position and, in whole mode, the following ones of the wave that lie in the same cell */
        const uint32_t pa = p0 + g0, oa = uni_u32(sidx[pa]), ha = uni_u32(shash[pa]);
        const int cx = (int)uni_u32((uint32_t)cell[3 * oa]), cy = (int)uni_u32((uint32_t)cell[3 * oa + 1]),
                  cz = (int)uni_u32((uint32_t)cell[3 * oa + 2]);
        uint32_t G = 1;
        if (whole)
            while (g0 + G < 4u && pa + G < un) {
                const uint32_t ob = uni_u32(sidx[pa + G]);
                if (uni_u32(shash[pa + G]) != ha || cell[3 * ob] != cx || cell[3 * ob + 1] != cy || cell[3 * ob + 2] != cz) break;
                G++;
            }
        float q0[4], q1[4], q2[4], q3[4], tx[4], ty[4], tz[4], lane_sum[4];
        uint32_t orig[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {                             /* members past the group's end repeat its last one; their sums are dropped */
            const uint32_t pk = pa + ((uint32_t)k < G ? (uint32_t)k : G - 1u);
            const float4 qq = sq[pk];
            q0[k] = qq.x; q1[k] = qq.y; q2[k] = qq.z; q3[k] = qq.w;
            tx[k] = st[3 * pk]; ty[k] = st[3 * pk + 1]; tz[k] = st[3 * pk + 2];
            orig[k] = sidx[pk];
            lane_sum[k] = 0.0f;
        }
        float votes = 1;                                          /* kernel.cu:722 (the sequential path: G == 1) */
        uint32_t first = 0, len = 0;
        if (lane < 27 && lane != 13) {
            const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
            const uint32_t h = fnv_cell3(cx + dx, cy + dy, cz + dz);
            if (h != 0u) {                                        /* a hash of 0 is never searched (kernel.cu:727) */
                for (uint32_t s = cell_slot(h, mask);; s = (s + 1u) & mask) {
                    const uint32_t k = tab[3u * s];
                    if (k == h) {
                        first = tab[3u * s + 1u];
                        const uint32_t end = tab[3u * s + 2u];
                        len = end > first && end <= un ? end - first : 0u;
                        break;
                    }
                    if (k == 0u) break;
                }
            }
        }
        uint32_t incl = len;                                      /* running sums over the 27 lanes */
        for (int o = 1; o < 32; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += up;
        }
        const uint32_t excl = incl - len, total = readlane_u(incl, 26);
        uint32_t c0 = 0;                                          /* scalar: the cell that holds the list position being placed */
        /* Four times 64 list positions per round, all their loads issued before the first test. */
        for (uint32_t base = 0; base < total; base += 4u * WAVE) {
            uint32_t j[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t b = base + (uint32_t)u * WAVE, f = b + (uint32_t)lane;
                in[u] = f < total;
                j[u] = 0;
                if (b >= total) continue;                         /* wave-uniform */
                while (readlane_u(incl, (int)c0) <= b) c0++;      /* b < total = incl[26]: stops at 26 at the latest */
                uint32_t cf, ce;
                if (b + WAVE <= readlane_u(incl, (int)c0)) {      /* the 64 positions lie in one cell: the usual case in a dense cluster */
                    cf = readlane_u(first, (int)c0);
                    ce = readlane_u(excl, (int)c0);
                } else {
                    uint32_t c = 0;                               /* the cell of list position f: running sum <= f */
#pragma unroll
                    for (uint32_t s2 = 16; s2 > 0; s2 >>= 1) {
                        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((c + s2 - 1u) << 2), (int)incl);
                        c += v <= f ? s2 : 0u;
                    }
                    c = c < 26u ? c : 26u;
                    cf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)first);
                    ce = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)excl);
                }
                j[u] = in[u] ? cf + (f - ce) : 0u;
            }
            float4 qo[4];
            float ex[4], ey[4], ez[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {                         /* position 0 for the lanes past the end: loaded, not used */
                qo[u] = sq[j[u]];
                ex[u] = st[3 * j[u]];
                ey[u] = st[3 * j[u] + 1];
                ez[u] = st[3 * j[u] + 2];
                w[u] = sw[j[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if ((uint32_t)k >= G) continue;               /* wave-uniform */
                    const float qd = fabsf(8 * (1 - (q0[k] * qo[u].x + q1[k] * qo[u].y + q2[k] * qo[u].z + q3[k] * qo[u].w)));
                    bool ok = in[u] && qd < rot_thresh_sq;
                    if (ok && !use_l1) {
                        /* sqrt(x) < d_dist with the correctly rounded, monotonic sqrt of the reference's sequence is
                         * x < dist2_below, the smallest float whose root reaches d_dist (made by the launcher): the same
                         * answer for every x, without a root per pair */
                        const float dx = tx[k] - ex[u], dy = ty[k] - ey[u], dz = tz[k] - ez[u];
                        ok = dx * dx + dy * dy + dz * dz < dist2_below;
                    }
                    if (whole) {
                        lane_sum[k] += ok ? w[u] : 0.0f;
                    } else {                                      /* G == 1 */
                        unsigned long long m = __ballot(ok);
                        while (m) {                               /* ascending list position: the reference's order */
                            const int b = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            votes += readlane_f(w[u], b);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k >= G) continue;
            float v = votes;
            if (whole) {
                float s = lane_sum[k];
                for (int o = WAVE / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
                v = 1 + s;
            }
            if (lane == 0) score[orig[k]] = v;
        }
        g0 += G;
    }
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
    hipLaunchKernelGGL((a->mode == 0 ? k_vote<0> : k_vote<1>), grid, dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    return oslamk_vote_wide(a, stream);             /* the redo list of this launch (oslam_vote_wide.hip) */
}

int oslamk_vote_group(const oslamk_vote_args *d_all, const oslamk_vote_args *h_all, int nm, void *stream)
{
    if (nm <= 0 || h_all[0].n_launch <= 0) return 0;
    int max_slices = 1;
    for (int j = 0; j < nm; j++) {
        if (h_all[j].n_launch != h_all[0].n_launch || h_all[j].mode != h_all[0].mode) return (int)hipErrorInvalidValue;
        if (h_all[j].table.n_slices > max_slices) max_slices = h_all[j].table.n_slices;
    }
    const size_t wgs = ((size_t)h_all[0].n_launch + 7) / 8 * 8 * (size_t)max_slices;
    if (wgs * (size_t)nm > 0x7fffffffu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((h_all[0].mode == 0 ? k_vote_group<0> : k_vote_group<1>), dim3((unsigned)(wgs * (size_t)nm)),
                       dim3(VOTE_THREADS), 0, (hipStream_t)stream, d_all, (uint32_t)wgs);
    return (int)hipGetLastError();      /* the caller runs oslamk_vote_wide for the members whose redo list is not empty */
}

size_t oslamk_cluster_table_words(int n)
{
    uint32_t cap = 64;
    while (cap < 2u * (uint32_t)(n > 0 ? n : 1)) cap <<= 1;
    return 3u * (size_t)cap;
}

int oslamk_cluster_scores(int n, const int *cell, const uint32_t *shash, const uint32_t *sidx,
                          const float *sq, const float *st, const float *sw, float d_dist, int use_l1,
                          float *score, int whole_host, const unsigned long long *whole_dev, uint32_t *table, void *stream)
{
    if (n <= 0) return 0;
    const size_t words = oslamk_cluster_table_words(n);
    const uint32_t mask = (uint32_t)(words / 3u) - 1u;
    hipError_t e = hipMemsetAsync(table, 0, sizeof(uint32_t) * words, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_cell_table_first, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, shash, n, table, mask);
    hipLaunchKernelGGL(k_cell_table_end, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, (hipStream_t)stream, shash, n, table, mask);
    /* the smallest float x with pm_sqrtf(x) >= d_dist: pm_sqrtf is correctly rounded, hence monotonic, so
     * pm_sqrtf(x) < d_dist  <=>  x < dist2_below for every float x (a NaN fails both) */
    float dist2_below = d_dist * d_dist;
    {
        uint32_t b;
        __builtin_memcpy(&b, &dist2_below, sizeof b);
        if (d_dist > 0.0f && b > 0x00800000u && b < 0x7f000000u) {
            float x = dist2_below;
            for (int it = 0; it < 8 && pm_sqrtf(x) >= d_dist; it++) { b--; __builtin_memcpy(&x, &b, sizeof x); }
            for (int it = 0; it < 16 && pm_sqrtf(x) < d_dist; it++) { b++; __builtin_memcpy(&x, &b, sizeof x); }
            dist2_below = x;
            /* checked, not assumed: the neighbours of the threshold fall on the right sides */
            uint32_t lo = b - 1u;
            float xl;
            __builtin_memcpy(&xl, &lo, sizeof xl);
            if (!(pm_sqrtf(x) >= d_dist) || !(pm_sqrtf(xl) < d_dist)) return (int)hipErrorInvalidValue;
        } else {
            return (int)hipErrorInvalidValue;       /* d_dist not a normal positive number: no clustering radius */
        }
    }
    hipLaunchKernelGGL(k_cluster_scores, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, (hipStream_t)stream, n, cell, shash, sidx,
                       table, mask, reinterpret_cast<const float4 *>(sq), st, sw, dist2_below, use_l1, score, whole_host, whole_dev);
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
