/*
 * oslam_sort.hip -- the hit sort of the PPF registration path (gfx950): per scene reference point, the hits
 * ordered by key number and the list of runs of equal keys that the vote kernel works from.
 *
 * The reference sorts its N^2 keys with thrust::sort_by_key (include/impl/parallel_hash_array.hpp:55-92) and looks
 * every scene pair up with thrust::lower_bound (model.cu:96-97).  Here a reference point has a few thousand hits,
 * each carrying the NUMBER of its key (16 bits for the 38 000 keys of a 5 k-point model, not the key's 32), and one
 * workgroup orders them in LDS with a hand-written stable counting sort on 8-bit digits, least significant first:
 *   per pass  1. every wave counts the digits of its contiguous share of the list (LDS atomics, 16 x 256 counters);
 *             2. the counters become write positions: digit-major, wave-minor prefix sums;
 *             3. every wave moves its share in order, 64 elements at a time: the lanes that hold the same digit
 *                find each other with eight ballots, take consecutive places from the digit's position and advance it.
 * An element is key number << 14 | position in the list (one 32-bit word: two buffers of 16 384 elements fit the LDS);
 * key numbers above 18 bits (databases of many models in one group) take 64-bit elements and half the segment.  The
 * run heads (a key changes, or 64 hits are full) come from the sorted words, the payloads are gathered into the second
 * list, and the sorted key numbers replace the arrival-order ones.  A list longer than a segment is sorted in segments (a key then has one run per segment it occurs in: its
 * bucket is streamed once per segment instead of once, everything else is unchanged).
 */
#include <hip/hip_runtime.h>

#include "oslam_kernels.h"
#include "ppf_core.h"

#define WAVE 64
#define SORT_THREADS 1024
#define SORT_WAVES (SORT_THREADS / WAVE)
#define SORT_IDX_BITS 14
#define SORT_SEG32 (1 << SORT_IDX_BITS)          /* hits per segment with 32-bit elements */
#define SORT_SEG64 (SORT_SEG32 / 2)              /* ... with 64-bit elements (key numbers above 18 bits) */
#define SORT_RADIX 256

__device__ __forceinline__ uint32_t sort_key_of(uint32_t e) { return e >> SORT_IDX_BITS; }
__device__ __forceinline__ uint32_t sort_idx_of(uint32_t e) { return e & (SORT_SEG32 - 1u); }
__device__ __forceinline__ uint32_t sort_pack(uint32_t key, uint32_t idx, uint32_t) { return (key << SORT_IDX_BITS) | idx; }
__device__ __forceinline__ uint32_t sort_key_of(unsigned long long e) { return (uint32_t)(e >> 32); }
__device__ __forceinline__ uint32_t sort_idx_of(unsigned long long e) { return (uint32_t)e; }
__device__ __forceinline__ unsigned long long sort_pack(uint32_t key, uint32_t idx, unsigned long long)
{
    return ((unsigned long long)key << 32) | idx;
}

/* One segment of n <= SEG hits: sort, gather, run heads.  bufA / bufB: SEG elements each; s_hist [SORT_WAVES][256];
 * s_part [SORT_WAVES + 1].  Returns the segment's run count (the same value in every thread). */
template <typename E, int SEG>
__device__ __forceinline__ uint32_t sort_segment(E *bufA, E *bufB, uint32_t *s_hist, uint32_t *s_part, const uint32_t *skey,
                                                 uint32_t *skey_out, const oslamk_pay *spay, oslamk_pay *dst, oslamk_run *runs, uint32_t n,
                                                 uint32_t seg, uint32_t n_runs, unsigned bits, int tid, int lane, int wid)
{
    for (uint32_t i = (uint32_t)tid; i < n; i += SORT_THREADS) bufA[i] = sort_pack(skey[i], i, E());
    /* the share of wave w: chunk hits, a multiple of 64, in list order */
    const uint32_t chunk = ((n + SORT_WAVES * WAVE - 1u) / (SORT_WAVES * WAVE)) * WAVE;
    const uint32_t c_lo = (uint32_t)wid * chunk < n ? (uint32_t)wid * chunk : n;
    const uint32_t c_hi = c_lo + chunk < n ? c_lo + chunk : n;
    uint32_t *hist = s_hist + wid * SORT_RADIX;
    E *src = bufA, *out = bufB;
    for (unsigned shift = 0; shift < bits; shift += 8) {
        for (int k = tid; k < SORT_WAVES * SORT_RADIX; k += SORT_THREADS) s_hist[k] = 0;
        __syncthreads();                /* also: the elements of the previous pass (or the load) are in src */
        for (uint32_t i = c_lo + (uint32_t)lane; i < c_hi; i += WAVE) atomicAdd(&hist[(sort_key_of(src[i]) >> shift) & 255u], 1u);
        __syncthreads();
        /* counters -> positions: everything with a smaller digit first, then the lower waves' share of the same digit */
        uint32_t tot = 0;
        if (tid < SORT_RADIX) {
            for (int w = 0; w < SORT_WAVES; w++) {
                const uint32_t v = s_hist[w * SORT_RADIX + tid];
                s_hist[w * SORT_RADIX + tid] = tot;
                tot += v;
            }
        }
        uint32_t incl = tot;            /* inclusive scan of the digit totals over the first four waves */
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += up;
        }
        if (tid < SORT_RADIX && lane == WAVE - 1) s_part[wid] = incl;
        __syncthreads();
        if (tid < SORT_RADIX) {
            uint32_t base = incl - tot;
            for (int w = 0; w < wid; w++) base += s_part[w];
            for (int w = 0; w < SORT_WAVES; w++) s_hist[w * SORT_RADIX + tid] += base;
        }
        __syncthreads();
        /* the move, in order */
        for (uint32_t i0 = c_lo; i0 < c_hi; i0 += WAVE) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool in = i < c_hi;
            const E e = src[in ? i : c_lo];
            const uint32_t digit = (sort_key_of(e) >> shift) & 255u;
            unsigned long long same = __ballot(in);            /* the lanes that hold the same digit as this one */
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bm = __ballot(in && ((digit >> b) & 1u));
                same &= (digit >> b) & 1u ? bm : ~bm;
            }
            const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if (in) {
                const uint32_t at = hist[digit];
                out[at + rank] = e;
                if (rank + 1u == (uint32_t)__popcll(same)) hist[digit] = at + rank + 1u;   /* the group's last lane moves the position on */
            }
        }
        __syncthreads();
        E *t = src;
        src = out;
        out = t;
    }
    /* src holds the sorted elements.  Every wave walks its share again, 64 positions at a time (a share starts on a
     * multiple of 64, and a run ends at every multiple of 64 hits: runs never leave their group of 64).  A head is a
     * position where the key changes or a group starts; first the heads are counted for the run numbers ... */
    uint32_t cnt = 0;
    for (uint32_t i0 = c_lo; i0 < c_hi; i0 += WAVE) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool in = i < c_hi;
        const bool head = in && (lane == 0 || sort_key_of(src[i]) != sort_key_of(src[i - 1u]));
        cnt += (uint32_t)__popcll(__ballot(head));
    }
    if (lane == 0) s_part[wid] = cnt;
    __syncthreads();
    uint32_t pos = n_runs, total = 0;
    for (int w = 0; w < SORT_WAVES; w++) {
        const uint32_t v = s_part[w];
        if (w < wid) pos += v;
        total += v;
    }
    /* ... then every head writes its run {key number | hits - 1, first hit | marker}: the hits of the run are the
     * lanes up to the next head, the marker is set when one of them carries it (rare: degenerate geometry); the
     * payloads go to the second list in sorted order (consecutive lanes, consecutive places) */
    for (uint32_t i0 = c_lo; i0 < c_hi; i0 += WAVE) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool in = i < c_hi;
        const E e = src[in ? i : c_lo];
        const bool head = in && (lane == 0 || sort_key_of(e) != sort_key_of(src[i - 1u]));
        oslamk_pay py;
        py.theta_t22 = 0;
        py.idx = 0;
        if (in) {
            py = spay[sort_idx_of(e)];
            dst[i] = py;
        }
        const unsigned long long hm = __ballot(head), im = __ballot(in), mm = __ballot(in && py.theta_t22 == PC_T22_FORCE);
        if (in) {
            /* the lanes of this lane's run: from the last head at or below it to the next head above it */
            const unsigned long long le = lane == WAVE - 1 ? ~0ull : (2ull << lane) - 1ull;
            const int start = 63 - __builtin_clzll(hm & le);
            const unsigned long long above = hm & ~le;
            const unsigned long long upto = above ? (1ull << (__ffsll((long long)above) - 1)) - 1ull : ~0ull;
            const unsigned long long mine = upto & ~((1ull << start) - 1ull) & im;
            const uint32_t marked = (mm & mine) ? 0x80000000u : 0u;
            /* the sorted key numbers, over the arrival-order ones this segment was loaded from (they are not read
             * again): what the vote kernel's near-edge search goes by; bit 31 = the hit's run holds a marker */
            skey_out[i] = sort_key_of(e) | marked;
            if (head) {
                oslamk_run rn;
                rn.slot_r = sort_key_of(e) | (((uint32_t)__popcll(mine) - 1u) << OSLAMK_RUN_SHIFT);
                rn.first = (seg + i) | marked;
                runs[pos + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = rn;
            }
        }
        pos += (uint32_t)__popcll(hm);
    }
    __syncthreads();                    /* src (LDS) has been read by everybody before the next segment overwrites it */
    return total;
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_hits(oslamk_vote_args a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[2 * SORT_SEG32];
    __shared__ uint32_t s_hist[SORT_WAVES * SORT_RADIX];
    __shared__ uint32_t s_part[SORT_WAVES + 1];
    const int ref_local = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const size_t off = a.hit_off[ref_local];
    const uint32_t cap = a.hit_off[ref_local + 1] - a.hit_off[ref_local];
    const uint32_t n_all = a.hit_count[ref_local] < cap ? a.hit_count[ref_local] : cap;     /* never past the list (oslamk_counters.list_overflow) */
    const unsigned bits = a.table.id_bits;
    const bool wide = bits + SORT_IDX_BITS > 32u;
    const uint32_t seg_max = wide ? SORT_SEG64 : SORT_SEG32;
    uint32_t n_runs = 0;                /* the same value in every thread */
    for (uint32_t seg = 0; seg < n_all; seg += seg_max) {
        const uint32_t n = n_all - seg < seg_max ? n_all - seg : seg_max;
        if (!wide)
            n_runs += sort_segment<uint32_t, SORT_SEG32>(s_buf, s_buf + SORT_SEG32, s_hist, s_part, a.hit_key + off + seg,
                                                         a.hit_key + off + seg, a.hit_pay + off + seg, a.hit_sorted + off + seg, a.runs + off, n, seg,
                                                         n_runs, bits, tid, lane, wid);
        else
            n_runs += sort_segment<unsigned long long, SORT_SEG64>(reinterpret_cast<unsigned long long *>(s_buf),
                                                                   reinterpret_cast<unsigned long long *>(s_buf) + SORT_SEG64, s_hist,
                                                                   s_part, a.hit_key + off + seg, a.hit_key + off + seg, a.hit_pay + off + seg,
                                                                   a.hit_sorted + off + seg, a.runs + off, n, seg, n_runs, bits, tid,
                                                                   lane, wid);
    }
    if (tid == 0) a.run_count[ref_local] = n_runs;
}

int oslamk_sort_hits(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    hipLaunchKernelGGL(k_sort_hits, dim3((unsigned)a->n_launch), dim3(SORT_THREADS), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
