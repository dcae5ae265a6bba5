/*
 * oslam_depth.hip -- depth image -> points + normals on the GPU: the front end of the streaming
 * configuration (a range camera in front of the PPF path; the reference's README.md:5-8 names
 * KinFu as the source of its scene clouds, it has no code for this step).  The statement this is
 * tested against, with every float operation in the same order, is oracle/oracle_depth.c.
 *
 * Per pixel (u, v) with depth z = raw * depth_scale, valid iff z_min <= z <= z_max:
 *   p = (((float)u - cx) * z / fx, ((float)v - cy) * z / fy, z)
 *   a normal needs the four axis neighbours valid and within max_jump of z (no normals across
 *   depth edges): n = cross(p(u+1,v) - p(u-1,v), p(u,v+1) - p(u,v-1)), normalised, turned to
 *   face the camera (n . p <= 0); a zero or non-finite cross product drops the pixel.
 * Output: the pixels that have a normal, in row-major pixel order (a stable compaction: flags ->
 * rocPRIM exclusive scan -> scatter), so the result does not depend on scheduling.
 */
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "oslam_kernels.h"
#include "ppf_math.h"

struct depth_cam {
    float fx, fy, cx, cy, scale, z_min, z_max, max_jump;
};

__device__ __forceinline__ float depth_at(const void *img, int is_u16, int w, int u, int v, float scale)
{
    const size_t i = (size_t)v * w + u;
    return is_u16 ? (float)reinterpret_cast<const uint16_t *>(img)[i] * scale
                  : reinterpret_cast<const float *>(img)[i] * scale;
}
__device__ __forceinline__ bool depth_ok(float z, const depth_cam &c) { return z >= c.z_min && z <= c.z_max; }
__device__ __forceinline__ void back_project(int u, int v, float z, const depth_cam &c, float p[3])
{
    p[0] = (((float)u - c.cx) * z) / c.fx;
    p[1] = (((float)v - c.cy) * z) / c.fy;
    p[2] = z;
}

/* one thread per pixel: out6[pixel] = x y z nx ny nz and flags[pixel] = 1 where the pixel has a normal */
__global__ __launch_bounds__(256) void k_depth_points(const void *img, int is_u16, int w, int h, depth_cam c,
                                                      float *out6, uint32_t *flags)
{
    const int u = blockIdx.x * 32 + (threadIdx.x & 31), v = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (u >= w || v >= h) return;
    const size_t i = (size_t)v * w + u;
    uint32_t ok = 0;
    const float z = depth_at(img, is_u16, w, u, v, c.scale);
    if (depth_ok(z, c) && u > 0 && v > 0 && u + 1 < w && v + 1 < h) {
        const float zl = depth_at(img, is_u16, w, u - 1, v, c.scale), zr = depth_at(img, is_u16, w, u + 1, v, c.scale);
        const float zu = depth_at(img, is_u16, w, u, v - 1, c.scale), zd = depth_at(img, is_u16, w, u, v + 1, c.scale);
        if (depth_ok(zl, c) && depth_ok(zr, c) && depth_ok(zu, c) && depth_ok(zd, c) &&
            pm_fabsf(zl - z) <= c.max_jump && pm_fabsf(zr - z) <= c.max_jump && pm_fabsf(zu - z) <= c.max_jump &&
            pm_fabsf(zd - z) <= c.max_jump) {
            float p[3], pl[3], pr[3], pu[3], pd[3];
            back_project(u, v, z, c, p);
            back_project(u - 1, v, zl, c, pl);
            back_project(u + 1, v, zr, c, pr);
            back_project(u, v - 1, zu, c, pu);
            back_project(u, v + 1, zd, c, pd);
            const float ax = pr[0] - pl[0], ay = pr[1] - pl[1], az = pr[2] - pl[2];
            const float bx = pd[0] - pu[0], by = pd[1] - pu[1], bz = pd[2] - pu[2];
            float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
            const float len = pm_sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0.0f && len <= 3.0e38f) {
                nx = nx / len;
                ny = ny / len;
                nz = nz / len;
                if (nx * p[0] + ny * p[1] + nz * p[2] > 0.0f) {
                    nx = -nx;
                    ny = -ny;
                    nz = -nz;
                }
                float *o = out6 + 6 * i;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
                o[3] = nx; o[4] = ny; o[5] = nz;
                ok = 1;
            }
        }
    }
    flags[i] = ok;
}

__global__ void k_depth_compact(const float *in6, const uint32_t *flags, const uint32_t *ord, size_t n, float *out6)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const float *s = in6 + 6 * i;
    float *o = out6 + 6 * (size_t)ord[i];
    for (int a = 0; a < 6; a++) o[a] = s[a];
}

/* [n][6] records -> structure of arrays (what the voxel-grid and key kernels read) */
__global__ void k_aos6_to_soa(const float *in6, size_t n, float *soa)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < 6; a++) soa[(size_t)a * n + i] = in6[6 * i + a];
}

extern "C" int oslamk_aos6_to_soa(const float *d_in6, size_t n, float *d_soa, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_aos6_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in6, n, d_soa);
    return (int)hipGetLastError();
}

#define DCHK(call)                   \
    do {                             \
        hipError_t e_ = (call);      \
        if (e_ != hipSuccess) {      \
            rc = (int)e_;            \
            goto done;               \
        }                            \
    } while (0)

/* d_img: the depth image in HBM (uint16 or float, row-major w x h).  d_out6: device [w*h][6].
 * Returns a hipError_t as int. */
extern "C" int oslamk_depth_to_cloud(const void *d_img, int is_u16, int w, int h, float fx, float fy, float cx,
                                     float cy, float scale, float z_min, float z_max, float max_jump, float *d_out6,
                                     uint32_t *n_out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = 0;
    const size_t n = (size_t)w * h;
    depth_cam c = {fx, fy, cx, cy, scale, z_min, z_max, max_jump};
    float *d_tmp6 = NULL;
    uint32_t *d_u = NULL;          /* flags, ord */
    void *d_scan = NULL;
    size_t scan_bytes = 0;
    uint32_t last[2] = {0, 0};
    *n_out = 0;
    if (n == 0) return 0;
    DCHK((hipError_t)oslam_dev_alloc((void **)&d_tmp6, sizeof(float) * 6 * n));
    DCHK((hipError_t)oslam_dev_alloc((void **)&d_u, sizeof(uint32_t) * 2 * n));
    hipLaunchKernelGGL(k_depth_points, dim3((w + 31) / 32, (h + 7) / 8), dim3(256), 0, stream, d_img, is_u16, w, h, c,
                       d_tmp6, d_u);
    DCHK(rocprim::exclusive_scan(nullptr, scan_bytes, d_u, d_u + n, 0u, n, rocprim::plus<uint32_t>(), stream));
    DCHK((hipError_t)oslam_dev_alloc(&d_scan, scan_bytes ? scan_bytes : 16));
    DCHK(rocprim::exclusive_scan(d_scan, scan_bytes, d_u, d_u + n, 0u, n, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_depth_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_tmp6, d_u, d_u + n, n,
                       d_out6);
    DCHK(hipMemcpyAsync(&last[0], d_u + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    DCHK(hipMemcpyAsync(&last[1], d_u + n + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    DCHK(hipStreamSynchronize(stream));
    DCHK(hipGetLastError());
    *n_out = last[0] + last[1];
done:
    oslam_dev_free(d_tmp6);
    oslam_dev_free(d_u);
    oslam_dev_free(d_scan);
    return rc;
}
