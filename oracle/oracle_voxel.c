/*
 * oracle_voxel.c -- CPU statement of the voxel-grid downsampling the reference applies to
 * every cloud before the PPF path (pcl/alignment/src/alignment.cpp:79-87,265-288 and
 * pcl/voxel_grid/voxel_grid.cpp:18-21: pcl::VoxelGrid, leaf = scene_leaf_size or d_dist).
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: PCL 1.7 is not vendored in the reference and not present here, and the
 * reference holds no fixture for this step.  What is restated is PCL's published algorithm
 * (pcl/filters/impl/voxel_grid.hpp, VoxelGrid<PointT>::applyFilter with its defaults:
 * downsample_all_data = true, min_points_per_voxel = 0, no field filter):
 *   1. bounding box of the finite points; min_b = floor(min * inv_leaf), max_b likewise,
 *      div_b = max_b - min_b + 1 (error if the voxel count overflows int32);
 *   2. voxel index of a point = sum_a (floor(p[a] * inv_leaf) - min_b[a]) * mul[a],
 *      mul = (1, div_b[0], div_b[0]*div_b[1]), inv_leaf = 1.0f / leaf in float;
 *   3. points sorted by voxel index; PCL uses std::sort (order inside a voxel unspecified),
 *      this statement keeps the original point order inside a voxel;
 *   4. one output point per occupied voxel, in ascending voxel index: every field (x, y, z,
 *      normal_x, normal_y, normal_z) is the float sum in that order divided by the count;
 *      normals are averaged, not renormalised (as PCL does).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint32_t idx; uint32_t pt; } vk;
static int vk_cmp(const void *a, const void *b)
{
    const vk *x = (const vk *)a, *y = (const vk *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

/* xyz, nrm: packed [n][3]; outputs packed, capacity n.  Returns the number of voxels, or -1. */
long orc_voxel_grid(const float *xyz, const float *nrm, size_t n, float leaf, float *xyz_out, float *nrm_out)
{
    const float inv = 1.0f / leaf;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int min_b[3], div_b[3];
    size_t i, m = 0, out = 0;
    vk *keys;
    for (i = 0; i < n; i++) {
        const float *p = xyz + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int a = 0; a < 3; a++) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; }
    }
    if (!(lo[0] <= hi[0])) return 0;
    for (int a = 0; a < 3; a++) {
        int64_t d = (int64_t)floorf(hi[a] * inv) - (int64_t)floorf(lo[a] * inv) + 1;
        min_b[a] = (int)floorf(lo[a] * inv);
        if (d > 0x7fffffff) return -1;
        div_b[a] = (int)d;
    }
    if ((int64_t)div_b[0] * div_b[1] > 0x7fffffff || (int64_t)div_b[0] * div_b[1] * div_b[2] > 0x7fffffff) return -1;
    keys = (vk *)malloc(sizeof(vk) * (n ? n : 1));
    for (i = 0; i < n; i++) {
        const float *p = xyz + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        keys[m].idx = (uint32_t)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
        keys[m].pt = (uint32_t)i;
        m++;
    }
    qsort(keys, m, sizeof(vk), vk_cmp);
    for (i = 0; i < m;) {
        size_t j = i;
        float s[6] = {0, 0, 0, 0, 0, 0};
        while (j < m && keys[j].idx == keys[i].idx) {
            for (int a = 0; a < 3; a++) { s[a] += xyz[3 * (size_t)keys[j].pt + a]; s[3 + a] += nrm[3 * (size_t)keys[j].pt + a]; }
            j++;
        }
        for (int a = 0; a < 3; a++) { xyz_out[3 * out + a] = s[a] / (float)(j - i); nrm_out[3 * out + a] = s[3 + a] / (float)(j - i); }
        out++;
        i = j;
    }
    free(keys);
    return (long)out;
}
