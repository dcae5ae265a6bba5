/*
 * oslam_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the PPF registration
 * path.  Wave = 64 lanes throughout.  Built with -ffp-contract=off: the float
 * sequences of ppf_core.h must round exactly as on the host.
 *
 * What the reference does with N^2-sized arrays and Thrust sorts
 * (pcl/alignment/src/cuda/{scene,model}.cu) is fused here:
 *   model build : pair key -> per-slice open-addressing table -> bucketed
 *                 16-byte pair entries (two counting passes, no sort);
 *   voting      : one workgroup per (scene reference point, model slice):
 *                 pair key -> table probe -> wave-cooperative sweep of the
 *                 bucket -> LDS accumulator [1024 model refs][32 alpha bins]
 *                 -> in-kernel peak extraction.
 */
#include <hip/hip_runtime.h>

#include "oslam_kernels.h"
#include "ppf_core.h"

#define WAVE 64
#define VOTE_THREADS 1024
#define ACC_CELLS (OSLAMK_SLICE * OSLAMK_NBIN)

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
    for (size_t i = b; i < e; i++) s += t.slots[i].len;
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
        run += t.slots[i].len;
    }
}

/* bit 31 of len: no lower slice holds this key (so hits are counted once) */
__global__ void k_table_mark_first(oslamk_table t, uint32_t *n_first)
{
    size_t total = (size_t)t.n_slices * t.cap;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    uint32_t key = t.slots[idx].key;
    if (key == 0) return;
    int slice = (int)(idx / t.cap);
    uint32_t mask = t.cap - 1;
    bool first = true;
    for (int s = 0; s < slice && first; s++) {
        const oslamk_slot *tab = t.slots + (size_t)s * t.cap;
        uint32_t slot = slot_of(key, t.shift);
        for (uint32_t probe = 0; probe < t.cap; probe++) {
            uint32_t k = tab[slot].key;
            if (k == key) { first = false; break; }
            if (k == 0) break;
            slot = (slot + 1) & mask;
        }
    }
    if (first) {
        t.slots[idx].len |= 0x80000000u;
        atomicAdd(n_first, 1u);
    }
}

/* pass 2: same pairs, written into their buckets */
__global__ void k_model_fill(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                             const float *tmg, oslamk_entry_exact *exact, oslamk_entry_fast *fast)
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
    uint32_t pos = atomicAdd(&tab[slot].cur, 1u);
    size_t e = (size_t)tab[slot].start + pos;
    const float *rows = tmg + 8 * (size_t)m_r;
    float uy = pc_row_dot(rows, x, y, z), uz = pc_row_dot(rows + 4, x, y, z);
    oslamk_entry_exact en;
    en.m_r = (uint32_t)m_r;
    en.uy = uy;
    en.uz = uz;
    en.m_i = (uint32_t)i;
    exact[e] = en;
    if (fast) {
        oslamk_entry_fast ef;
        ef.m_r = (uint32_t)m_r;
        ef.am = (pm_atan2f(uz, uy) + PM_PI_F) / PM_D_ANGLE;
        fast[e] = ef;
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

/* One workgroup = one (scene reference point, model slice).
 * LDS: acc[1024][32] u32 = 128 KiB (one workgroup per CU, 16 waves).
 * ComputeUniqueVotes (model.cu:95-171) without the vote list: K3/K4
 * (kernel.cu:480-554) accumulate straight into acc, and the sort/histogram/
 * threshold of model.cu:148-170 becomes the scan at the end. */
template <int MODE>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote(oslamk_vote_args a)
{
    __shared__ uint32_t acc[ACC_CELLS];
    __shared__ uint32_t s_wave[VOTE_THREADS / WAVE];
    __shared__ uint32_t s_wave2[VOTE_THREADS / WAVE];
    __shared__ unsigned long long s_wave64[2][VOTE_THREADS / WAVE];
    __shared__ uint32_t s_g, s_lmax, s_base;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int n_slices = a.table.n_slices;
    const int ref_ord = a.first_ref + (int)(blockIdx.x / n_slices);
    const int slice = (int)(blockIdx.x % n_slices);
    const int S = a.scene.n;
    const uint32_t r = a.ref_idx[ref_ord];

    for (int c = tid; c < ACC_CELLS; c += VOTE_THREADS) acc[c] = 0;

    const float prx = a.scene.px[r], pry = a.scene.py[r], prz = a.scene.pz[r];
    const float nrx = a.scene.nx[r], nry = a.scene.ny[r], nrz = a.scene.nz[r];
    const float nrn = pc_norm3(nrx, nry, nrz);
    float ty[4], tz[4];
    {
        const float *rows = a.tsg + 8 * (size_t)ref_ord;
        for (int q = 0; q < 4; q++) { ty[q] = rows[q]; tz[q] = rows[4 + q]; }
    }
    const oslamk_slot *tab = a.table.slots + (size_t)slice * a.table.cap;
    const uint32_t mask = a.table.cap - 1, shift = a.table.shift;
    const uint32_t m_base = (uint32_t)slice * OSLAMK_SLICE;
    unsigned long long my_hits = 0, my_votes = 0;
    __syncthreads();

    for (int base = 0; base < S; base += VOTE_THREADS) {
        const int i = base + tid;
        uint32_t start = 0, len = 0;
        float vy = 0.0f, vz = 0.0f;
        if (i < S && (uint32_t)i != r) {
            float x, y, z;
            uint32_t key = cloud_pair_key(a.scene, i, prx, pry, prz, nrx, nry, nrz, nrn, a.d_dist,
                                          a.inv_d_dist, &x, &y, &z);
            if (key != 0) {                                   /* kernel.cu:491,520 */
                uint32_t slot = slot_of(key, shift);
                for (uint32_t probe = 0; probe <= mask; probe++) {
                    const uint4 sv = *reinterpret_cast<const uint4 *>(&tab[slot]);
                    if (sv.x == key) {
                        start = sv.y;
                        len = sv.z & 0x7fffffffu;
                        my_hits += sv.z >> 31;
                        my_votes += len;
                        break;
                    }
                    if (sv.x == 0) break;
                    slot = (slot + 1) & mask;
                }
                if (len) {
                    vy = pc_row_dot(ty, x, y, z);            /* kernel.cu:334-336 */
                    vz = pc_row_dot(tz, x, y, z);
                    if (MODE == 1) {
                        /* fast mode: vy carries (alpha_s + pi)/D + 45 */
                        vy = (pm_atan2f(vz, vy) + PM_PI_F) / PM_D_ANGLE + 45.0f;
                    }
                }
            }
        }
        /* wave-cooperative sweep: all 64 lanes stream one bucket at a time */
        unsigned long long todo = __ballot(len > 0);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t st = readlane_u(start, l), ln = readlane_u(len, l);
            const float vyl = readlane_f(vy, l), vzl = readlane_f(vz, l);
            if (MODE == 0) {
                for (uint32_t e = lane; e < ln; e += WAVE) {
                    const uint4 ev = *reinterpret_cast<const uint4 *>(&a.exact[(size_t)st + e]);
                    const unsigned bin = pc_alpha_bin_exact(__builtin_bit_cast(float, ev.y),
                                                            __builtin_bit_cast(float, ev.z), vyl, vzl);
                    if (bin < OSLAMK_NBIN) atomicAdd(&acc[(ev.x - m_base) * OSLAMK_NBIN + bin], 1u);
                }
            } else {
                for (uint32_t e = lane; e < ln; e += WAVE) {
                    const uint2 ev = *reinterpret_cast<const uint2 *>(&a.fast[(size_t)st + e]);
                    float t = vyl - __builtin_bit_cast(float, ev.y);   /* in [15, 75] */
                    t = t >= 30.0f ? t - 30.0f : t;
                    t = t >= 30.0f ? t - 30.0f : t;
                    const unsigned bin = (unsigned)(int)t;
                    if (bin < OSLAMK_NBIN) atomicAdd(&acc[(ev.x - m_base) * OSLAMK_NBIN + bin], 1u);
                }
            }
        }
    }
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
 * device self-test of the float path
 * ------------------------------------------------------------------------*/
__global__ void k_selftest(const float *x, const float *y, const float *x2, size_t n, float *out_acos,
                           float *out_atan2, uint32_t *out_quant, uint32_t *out_bin)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_acos[i] = pm_acosf(x[i]);
    out_atan2[i] = pm_atan2f(y[i], x2[i]);
    out_quant[i] = pc_quant_bits(pm_fabsf(y[i]) * 7.0f, 0.0371f + pm_fabsf(x[i]) * 0.01f,
                                 1.0f / (0.0371f + pm_fabsf(x[i]) * 0.01f));
    out_bin[i] = pc_alpha_bin_exact(y[i], x2[i], x[i], y[i] - x2[i]);
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

int oslamk_table_mark_first(oslamk_table t, uint32_t *n_first, void *stream)
{
    size_t total = (size_t)t.n_slices * t.cap;
    hipLaunchKernelGGL(k_table_mark_first, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, t, n_first);
    return (int)hipGetLastError();
}

int oslamk_model_fill(oslamk_cloud c, float d_dist, float inv_d_dist, oslamk_table t,
                      const float *tmg, oslamk_entry_exact *exact, oslamk_entry_fast *fast,
                      void *stream)
{
    hipLaunchKernelGGL(k_model_fill, dim3((c.n + 255) / 256, c.n), dim3(256), 0, (hipStream_t)stream,
                       c, d_dist, inv_d_dist, t, tmg, exact, fast);
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

int oslamk_selftest(const float *x, const float *y, const float *x2, size_t n, float *out_acos,
                    float *out_atan2, uint32_t *out_quant, uint32_t *out_bin, void *stream)
{
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, x2, n, out_acos, out_atan2, out_quant, out_bin);
    return (int)hipGetLastError();
}

} /* extern "C" */
