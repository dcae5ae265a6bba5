/*
 * oslam_voxel.hip -- voxel-grid downsampling on the GPU: the step the reference runs on every
 * cloud right before the PPF path (pcl/alignment/src/alignment.cpp:79-87,265-288: pcl::VoxelGrid
 * with leaf = scene_leaf_size for scenes and leaf = d_dist for models; pcl/voxel_grid/
 * voxel_grid.cpp:18-21).  Semantics: PCL's VoxelGrid<PointT>::applyFilter with its defaults, with
 * the one freedom PCL leaves (order of points inside a voxel, std::sort) fixed to the original
 * point order -- see oracle/oracle_voxel.c for the statement this is tested against.
 *
 * Pipeline: bounding box (block reduce) -> voxel index per point -> rocPRIM radix sort of
 * (index, point) pairs (stable) -> run heads + exclusive scan -> one thread per occupied voxel
 * sums its run in order (float, sequential: deterministic) and divides by the count.
 */
#include <hip/hip_runtime.h>

#include <cstring>   /* rocPRIM's texture iterator calls memset without including it */

#include <rocprim/rocprim.hpp>

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "oslam_kernels.h"

#define VOX_INVALID 0xffffffffu

__global__ __launch_bounds__(256) void k_vox_bbox(oslamk_cloud c, float *blk_lo, float *blk_hi)
{
    __shared__ float s_lo[3][256], s_hi[3][256];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += gridDim.x * blockDim.x) {
        const float p[3] = {c.px[i], c.py[i], c.pz[i]};
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
    for (int a = 0; a < 3; a++) { s_lo[a][threadIdx.x] = lo[a]; s_hi[a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; a++) {
                s_lo[a][threadIdx.x] = fminf(s_lo[a][threadIdx.x], s_lo[a][threadIdx.x + s]);
                s_hi[a][threadIdx.x] = fmaxf(s_hi[a][threadIdx.x], s_hi[a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        blk_lo[3 * blockIdx.x + threadIdx.x] = s_lo[threadIdx.x][0];
        blk_hi[3 * blockIdx.x + threadIdx.x] = s_hi[threadIdx.x][0];
    }
}

__global__ void k_vox_index(oslamk_cloud c, float inv, int mb0, int mb1, int mb2, int d0, int d01,
                            uint32_t *keys, uint32_t *vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    const float x = c.px[i], y = c.py[i], z = c.pz[i];
    uint32_t k = VOX_INVALID;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        const int i0 = (int)(floorf(x * inv) - (float)mb0);
        const int i1 = (int)(floorf(y * inv) - (float)mb1);
        const int i2 = (int)(floorf(z * inv) - (float)mb2);
        k = (uint32_t)(i0 + i1 * d0 + i2 * d01);
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
}

__global__ void k_vox_heads(const uint32_t *keys, int n, uint32_t *flags)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    flags[k] = (keys[k] != VOX_INVALID && (k == 0 || keys[k] != keys[k - 1])) ? 1u : 0u;
}

__global__ void k_vox_average(oslamk_cloud c, const uint32_t *keys, const uint32_t *vals,
                              const uint32_t *flags, const uint32_t *ord, int n, float *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || !flags[k]) return;
    const uint32_t key = keys[k];
    float s[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int j = k;
    for (; j < n && keys[j] == key; j++) {
        const uint32_t p = vals[j];
        s[0] += c.px[p]; s[1] += c.py[p]; s[2] += c.pz[p];
        s[3] += c.nx[p]; s[4] += c.ny[p]; s[5] += c.nz[p];
    }
    const float cnt = (float)(j - k);
    float *o = out + 6 * (size_t)ord[k];
    for (int a = 0; a < 6; a++) o[a] = s[a] / cnt;
}

#define VCHK(call)                   \
    do {                             \
        hipError_t e_ = (call);      \
        if (e_ != hipSuccess) {      \
            rc = (int)e_;            \
            goto done;               \
        }                            \
    } while (0)

/* c: cloud in HBM (SoA).  out6: device buffer [n][6] (x y z nx ny nz per voxel).
 * Returns a hipError_t as int; -1: leaf too small (voxel count overflows int32). */
extern "C" int oslamk_voxel_grid(oslamk_cloud c, float leaf, float *out6, uint32_t *n_out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = 0;
    const int n = c.n, nblk = n < 256 * 1024 ? (n + 255) / 256 : 1024;
    const float inv = 1.0f / leaf;
    float *d_lo = NULL, *d_hi = NULL, *h = NULL;
    uint32_t *d_u = NULL;       /* keys_in, vals_in, keys_out, vals_out, flags, ord : 6*n */
    void *d_tmp = NULL;
    size_t tmp_sort = 0, tmp_scan = 0, tmp_bytes;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int mb[3], db[3];
    uint32_t last[2] = {0, 0};
    *n_out = 0;
    if (n <= 0) return 0;
    VCHK((hipError_t)oslam_dev_alloc((void **)&d_lo, sizeof(float) * 3 * nblk));
    VCHK((hipError_t)oslam_dev_alloc((void **)&d_hi, sizeof(float) * 3 * nblk));
    hipLaunchKernelGGL(k_vox_bbox, dim3(nblk), dim3(256), 0, stream, c, d_lo, d_hi);
    h = (float *)malloc(sizeof(float) * 6 * nblk);
    VCHK(hipMemcpyAsync(h, d_lo, sizeof(float) * 3 * nblk, hipMemcpyDeviceToHost, stream));
    VCHK(hipMemcpyAsync(h + 3 * nblk, d_hi, sizeof(float) * 3 * nblk, hipMemcpyDeviceToHost, stream));
    VCHK(hipStreamSynchronize(stream));
    for (int b = 0; b < nblk; b++)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], h[3 * b + a]);
            hi[a] = fmaxf(hi[a], h[3 * nblk + 3 * b + a]);
        }
    if (!(lo[0] <= hi[0])) goto done;            /* no finite point */
    for (int a = 0; a < 3; a++) {
        const long long d = (long long)floorf(hi[a] * inv) - (long long)floorf(lo[a] * inv) + 1;
        mb[a] = (int)floorf(lo[a] * inv);
        if (d > 0x7fffffffLL) { rc = -1; goto done; }
        db[a] = (int)d;
    }
    if ((long long)db[0] * db[1] > 0x7fffffffLL || (long long)db[0] * db[1] * db[2] > 0x7fffffffLL) { rc = -1; goto done; }

    VCHK((hipError_t)oslam_dev_alloc((void **)&d_u, sizeof(uint32_t) * 6 * (size_t)n));
    {
        uint32_t *k_in = d_u, *v_in = d_u + n, *k_out = d_u + 2 * (size_t)n, *v_out = d_u + 3 * (size_t)n;
        uint32_t *flags = d_u + 4 * (size_t)n, *ord = d_u + 5 * (size_t)n;
        hipLaunchKernelGGL(k_vox_index, dim3((n + 255) / 256), dim3(256), 0, stream, c, inv, mb[0], mb[1], mb[2],
                           db[0], db[0] * db[1], k_in, v_in);
        VCHK(rocprim::radix_sort_pairs(nullptr, tmp_sort, k_in, k_out, v_in, v_out, (size_t)n, 0, 32, stream));
        VCHK(rocprim::exclusive_scan(nullptr, tmp_scan, flags, ord, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
        tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
        VCHK((hipError_t)oslam_dev_alloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
        VCHK(rocprim::radix_sort_pairs(d_tmp, tmp_sort, k_in, k_out, v_in, v_out, (size_t)n, 0, 32, stream));
        hipLaunchKernelGGL(k_vox_heads, dim3((n + 255) / 256), dim3(256), 0, stream, k_out, n, flags);
        VCHK(rocprim::exclusive_scan(d_tmp, tmp_scan, flags, ord, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_vox_average, dim3((n + 255) / 256), dim3(256), 0, stream, c, k_out, v_out, flags, ord, n, out6);
        VCHK(hipMemcpyAsync(&last[0], flags + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        VCHK(hipMemcpyAsync(&last[1], ord + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        VCHK(hipStreamSynchronize(stream));
        VCHK(hipGetLastError());
        *n_out = last[0] + last[1];
    }
done:
    free(h);
    oslam_dev_free(d_lo);
    oslam_dev_free(d_hi);
    oslam_dev_free(d_u);
    oslam_dev_free(d_tmp);
    return rc;
}
