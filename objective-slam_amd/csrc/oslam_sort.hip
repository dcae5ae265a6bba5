/*
 * oslam_sort.hip -- the hit sort of the PPF registration path (gfx950): per scene reference point, the hits
 * ordered by key number and the list of runs of equal keys that the vote kernel works from.
 *
 * A translation unit of its own because it is built with -mllvm -disable-machine-licm: with the pass on, the
 * compiler hoists enough loop invariants of rocPRIM's sort over its passes to run the kernel, which has 16 keys and
 * 16 values per thread in registers, 4 registers over its 128 (20 bytes of scratch per lane); without it the
 * kernel fits (profiles/r02_kernel_resources.txt) at the same speed.
 */
#include <hip/hip_runtime.h>

#include <cstring>   /* rocPRIM calls memset without including it */

#include <rocprim/block/block_radix_sort.hpp>

#include "oslam_kernels.h"
#include "ppf_core.h"

#define WAVE 64

/* Orders the hit list of each reference point of the batch by key (by the key's slot in the union
 * table: log2(ucap) bits instead of 32), so that hits that share a bucket are adjacent (on the bench
 * scene a bucket is hit 3.8 times per reference point on average; streaming it once per run cuts
 * the entry traffic 4.7x), and writes the run list the vote kernel works from:
 * runs[u] = {slot | (hits - 1) << 26, index of the run's first hit | marker << 31}; a run also ends
 * at every multiple of 64 hits, so a run is at most one hit per lane.
 * One workgroup per reference point: radix sort of (slot, index) in LDS -- rocPRIM's block
 * primitive (DESIGN.md 4 says why it stays) -- then the payloads are gathered into the second list.
 * A list longer than SORT_MAX is sorted in segments of SORT_MAX hits (a key then has one run per
 * segment it occurs in: its bucket is streamed once per segment instead of once, everything else
 * is unchanged). */
#define SORT_MAX 16384
#define SORT_THREADS 1024
#define SORT_ITEMS (SORT_MAX / SORT_THREADS)
#define SORT_SMALL_ITEMS 2                      /* lists of up to 2048 hits (small scenes) take a 2-per-thread sort */
typedef rocprim::block_radix_sort<uint32_t, SORT_THREADS, SORT_ITEMS, uint32_t> hit_block_sort;
typedef rocprim::block_radix_sort<uint32_t, SORT_THREADS, SORT_SMALL_ITEMS, uint32_t> hit_block_sort_small;

/* One segment of n <= ITEMS * 1024 hits: sort, gather, run heads.  Returns the segment's run count (the
 * same value in every thread). */
template <int ITEMS, class SORT>
__device__ __forceinline__ uint32_t sort_segment(typename SORT::storage_type &s_sort, uint32_t *s_part, uint32_t *s_last,
                                                 const uint32_t *skey, const oslamk_pay *spay, oslamk_pay *dst,
                                                 oslamk_run *runs, uint32_t n, uint32_t seg, uint32_t n_runs,
                                                 unsigned bits, int tid, int lane, int wid)
{
    /* Thread t brings the hits ITEMS*t .. ITEMS*t+ITEMS-1 and ends up with the same sorted positions in
     * registers.  Places past the end carry all ones; the sort is stable and they come last in the
     * input order, so they also come last among equal keys and the first n sorted positions are the
     * hits. */
    const uint32_t i0 = (uint32_t)tid * ITEMS;
    uint32_t key[ITEMS], idx[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = i0 + k;
        key[k] = i < n ? skey[i] : 0xffffffffu;
        idx[k] = i;
    }
    SORT().sort(key, idx, s_sort, 0u, bits);
    uint32_t marked = 0;                 /* sorted positions of this thread whose hit carries the marker */
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
        if (i0 + k < n) {
            const oslamk_pay py = spay[idx[k]];
            dst[i0 + k] = py;
            marked |= (uint32_t)(py.theta_t22 == PC_T22_FORCE) << k;
        }
    s_last[tid] = key[ITEMS - 1];
    __syncthreads();
    uint32_t heads = 0, cnt = 0;
    uint32_t prev = tid ? s_last[tid - 1] : 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = i0 + k;
        if (i < n) {
            const bool head = (i & (WAVE - 1)) == 0 || key[k] != prev;
            heads |= (uint32_t)head << k;
            cnt += head;
        }
        prev = key[k];
    }
    uint32_t incl = cnt;
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += up;
    }
    if (lane == WAVE - 1) s_part[wid] = incl;
    __syncthreads();
    uint32_t pos = n_runs + incl - cnt, total = 0;
    for (int w = 0; w < SORT_THREADS / WAVE; w++) {
        const uint32_t v = s_part[w];
        if (w < wid) pos += v;
        total += v;
    }
    const uint32_t pos0 = pos;          /* index of this thread's first head; the run before it holds its leading positions */
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
        if ((heads >> k) & 1u) {
            oslamk_run rn;
            rn.slot_r = key[k];
            rn.first = seg + i0 + k;
            runs[pos++] = rn;
        }
    __threadfence_block();
    __syncthreads();                    /* the run heads are written; the LDS arrays are reused by the next segment */
    /* bit 31 of `first`: some hit of the run carries the marker (rare: degenerate geometry) */
    if (marked) {
        uint32_t ri = pos0 - 1u;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            ri += (heads >> k) & 1u;
            if ((marked >> k) & 1u) atomicOr(&runs[ri].first, 0x80000000u);
        }
    }
    /* length of every run = distance to the next head (or to the end of the segment) */
    for (uint32_t u = (uint32_t)tid; u < total; u += SORT_THREADS) {
        const uint32_t f0 = runs[n_runs + u].first & 0x7fffffffu;
        const uint32_t f1 = u + 1 < total ? runs[n_runs + u + 1].first & 0x7fffffffu : seg + n;
        runs[n_runs + u].slot_r |= (f1 - f0 - 1u) << OSLAMK_RUN_SHIFT;
    }
    return total;
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_hits(oslamk_vote_args a)
{
    __shared__ union {
        typename hit_block_sort::storage_type big;
        typename hit_block_sort_small::storage_type small;
    } s_sort;
    __shared__ uint32_t s_part[SORT_THREADS / WAVE];
    __shared__ uint32_t s_last[SORT_THREADS];           /* the last key of every thread's sorted positions */
    const int ref_local = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const size_t off = a.hit_off[ref_local];
    const uint32_t cap = a.hit_off[ref_local + 1] - a.hit_off[ref_local];
    const uint32_t n_all = a.hit_count[ref_local] < cap ? a.hit_count[ref_local] : cap;     /* never past the list (oslamk_counters.list_overflow) */
    const unsigned bits = a.table.id_bits;
    uint32_t n_runs = 0;                /* the same value in every thread */
    for (uint32_t seg = 0; seg < n_all; seg += SORT_MAX) {
        const uint32_t n = n_all - seg < SORT_MAX ? n_all - seg : SORT_MAX;
        if (n <= SORT_SMALL_ITEMS * SORT_THREADS)
            n_runs += sort_segment<SORT_SMALL_ITEMS, hit_block_sort_small>(s_sort.small, s_part, s_last, a.hit_key + off + seg,
                                                                           a.hit_pay + off + seg, a.hit_sorted + off + seg,
                                                                           a.runs + off, n, seg, n_runs, bits, tid, lane, wid);
        else
            n_runs += sort_segment<SORT_ITEMS, hit_block_sort>(s_sort.big, s_part, s_last, a.hit_key + off + seg,
                                                               a.hit_pay + off + seg, a.hit_sorted + off + seg, a.runs + off, n,
                                                               seg, n_runs, bits, tid, lane, wid);
    }
    if (tid == 0) a.run_count[ref_local] = n_runs;
}

int oslamk_sort_hits(const oslamk_vote_args *a, void *stream)
{
    if (a->n_launch <= 0) return 0;
    hipLaunchKernelGGL(k_sort_hits, dim3((unsigned)a->n_launch), dim3(1024), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
