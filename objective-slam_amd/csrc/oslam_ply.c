/*
 * oslam_ply.c -- PLY point clouds with normals, in C: what the reference gets from
 * pcl::io::loadPLYFile<pcl::PointNormal> (pcl/alignment/src/alignment.cpp:212,241) and writes
 * with pcl::PLYWriter (pcl/voxel_grid/voxel_grid.cpp:27-29) or matlab/write_ply_cloud.m.
 * Host-only.  Reads `format ascii 1.0` and `format binary_little_endian 1.0`; the vertex
 * element must come first; it needs x, y, z and normals named nx ny nz or
 * normal_x normal_y normal_z (any scalar type, any other properties are skipped).
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oslam.h"

static int type_size(const char *t)
{
    if (!strcmp(t, "char") || !strcmp(t, "uchar") || !strcmp(t, "int8") || !strcmp(t, "uint8")) return 1;
    if (!strcmp(t, "short") || !strcmp(t, "ushort") || !strcmp(t, "int16") || !strcmp(t, "uint16")) return 2;
    if (!strcmp(t, "int") || !strcmp(t, "uint") || !strcmp(t, "float") || !strcmp(t, "int32") ||
        !strcmp(t, "uint32") || !strcmp(t, "float32")) return 4;
    if (!strcmp(t, "double") || !strcmp(t, "float64")) return 8;
    return 0;
}

static double read_scalar(const unsigned char *p, const char *t)
{
    if (!strcmp(t, "float") || !strcmp(t, "float32")) { float v; memcpy(&v, p, 4); return v; }
    if (!strcmp(t, "double") || !strcmp(t, "float64")) { double v; memcpy(&v, p, 8); return v; }
    if (!strcmp(t, "char") || !strcmp(t, "int8")) return (signed char)p[0];
    if (!strcmp(t, "uchar") || !strcmp(t, "uint8")) return p[0];
    if (!strcmp(t, "short") || !strcmp(t, "int16")) { short v; memcpy(&v, p, 2); return v; }
    if (!strcmp(t, "ushort") || !strcmp(t, "uint16")) { unsigned short v; memcpy(&v, p, 2); return v; }
    if (!strcmp(t, "int") || !strcmp(t, "int32")) { int v; memcpy(&v, p, 4); return v; }
    { unsigned v; memcpy(&v, p, 4); return v; }
}

#define MAX_PROPS 64

int oslam_ply_read(const char *path, float **xyz_out, float **nrm_out, size_t *n_out)
{
    FILE *f;
    char line[1024], types[MAX_PROPS][16];
    int nprop = 0, ascii = -1, in_vertex = 0, seen_vertex = 0, off[MAX_PROPS], slot[MAX_PROPS], rec = 0, i;
    int have[6] = {0, 0, 0, 0, 0, 0};
    size_t n = 0, v;
    float *xyz = NULL, *nrm = NULL;
    unsigned char *buf = NULL;
    int rc = OSLAM_E_INVALID;
    if (!path || !xyz_out || !nrm_out || !n_out) return OSLAM_E_INVALID;
    *xyz_out = *nrm_out = NULL;
    *n_out = 0;
    f = fopen(path, "rb");
    if (!f) return OSLAM_E_INVALID;
    if (!fgets(line, sizeof line, f) || strncmp(line, "ply", 3)) goto done;
    while (fgets(line, sizeof line, f)) {
        char a[64] = "", b[64] = "", c[64] = "";
        int k = sscanf(line, "%63s %63s %63s", a, b, c);
        if (k < 1) continue;
        if (!strcmp(a, "end_header")) break;
        if (!strcmp(a, "format")) {
            if (!strcmp(b, "ascii")) ascii = 1;
            else if (!strcmp(b, "binary_little_endian")) ascii = 0;
            else goto done;                                   /* big endian: not supported */
        } else if (!strcmp(a, "element")) {
            in_vertex = !strcmp(b, "vertex");
            if (in_vertex) {
                if (seen_vertex) goto done;
                seen_vertex = 1;
                n = (size_t)strtoull(c, NULL, 10);
            } else if (!seen_vertex) {
                goto done;                                    /* an element before the vertices */
            }
        } else if (!strcmp(a, "property") && in_vertex) {
            static const char *names[6][2] = {{"x", "x"}, {"y", "y"}, {"z", "z"}, {"nx", "normal_x"},
                                              {"ny", "normal_y"}, {"nz", "normal_z"}};
            if (!strcmp(b, "list") || nprop == MAX_PROPS || !type_size(b)) goto done;
            snprintf(types[nprop], sizeof types[nprop], "%.15s", b);
            off[nprop] = rec;
            rec += type_size(b);
            slot[nprop] = -1;
            for (i = 0; i < 6; i++)
                if (!strcmp(c, names[i][0]) || !strcmp(c, names[i][1])) { slot[nprop] = i; have[i] = 1; }
            nprop++;
        }
    }
    if (ascii < 0 || !seen_vertex || !(have[0] && have[1] && have[2] && have[3] && have[4] && have[5])) goto done;
    xyz = (float *)malloc(sizeof(float) * 3 * (n ? n : 1));
    nrm = (float *)malloc(sizeof(float) * 3 * (n ? n : 1));
    if (!xyz || !nrm) { rc = OSLAM_E_NOMEM; goto done; }
    if (ascii) {
        for (v = 0; v < n; v++) {
            for (i = 0; i < nprop; i++) {
                double d;
                if (fscanf(f, "%lf", &d) != 1) goto done;
                if (slot[i] >= 0 && slot[i] < 3) xyz[3 * v + slot[i]] = (float)d;
                else if (slot[i] >= 3) nrm[3 * v + slot[i] - 3] = (float)d;
            }
        }
    } else {
        buf = (unsigned char *)malloc((size_t)rec * (n ? n : 1));
        if (!buf) { rc = OSLAM_E_NOMEM; goto done; }
        if (fread(buf, (size_t)rec, n, f) != n) goto done;
        for (v = 0; v < n; v++)
            for (i = 0; i < nprop; i++) {
                if (slot[i] < 0) continue;
                float d = (float)read_scalar(buf + v * (size_t)rec + off[i], types[i]);
                if (slot[i] < 3) xyz[3 * v + slot[i]] = d; else nrm[3 * v + slot[i] - 3] = d;
            }
    }
    *xyz_out = xyz;
    *nrm_out = nrm;
    *n_out = n;
    xyz = nrm = NULL;
    rc = OSLAM_OK;
done:
    free(xyz);
    free(nrm);
    free(buf);
    fclose(f);
    return rc;
}

int oslam_ply_write(const char *path, const float *xyz, const float *nrm, size_t n, int binary)
{
    FILE *f;
    size_t v;
    if (!path || !xyz || !nrm) return OSLAM_E_INVALID;
    f = fopen(path, "wb");
    if (!f) return OSLAM_E_INVALID;
    fprintf(f, "ply\nformat %s 1.0\ncomment written by liboslam_hip\nelement vertex %zu\n"
               "property float x\nproperty float y\nproperty float z\n"
               "property float normal_x\nproperty float normal_y\nproperty float normal_z\nend_header\n",
            binary ? "binary_little_endian" : "ascii", n);
    for (v = 0; v < n; v++) {
        if (binary) {
            fwrite(xyz + 3 * v, sizeof(float), 3, f);
            fwrite(nrm + 3 * v, sizeof(float), 3, f);
        } else {
            fprintf(f, "%.9g %.9g %.9g %.9g %.9g %.9g\n", xyz[3 * v], xyz[3 * v + 1], xyz[3 * v + 2], nrm[3 * v],
                    nrm[3 * v + 1], nrm[3 * v + 2]);
        }
    }
    return fclose(f) ? OSLAM_E_INVALID : OSLAM_OK;
}

void oslam_free(void *p) { free(p); }
