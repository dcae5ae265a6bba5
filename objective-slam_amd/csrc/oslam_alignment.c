/*
 * oslam_alignment.c -- command-line front end with the flags of the reference's `alignment`
 * program (pcl/alignment/src/alignment.cpp:119-172), minus the visualiser:
 *
 *   oslam_alignment --scene_files a.ply,b.ply --model_files m.ply --tau_d 0.05 \
 *       [--scene_leaf_size 10.0] [--ref_point_df 1] [--vote_count_threshold 0.4]
 *       [--cpu_clustering false] [--use_l1_norm false] [--use_averaged_clusters false]
 *       [--validation_files truth.txt] [--validation_translation_threshold 0.1]
 *       [--validation_rotation_threshold 12] [--dev 1]
 *
 * Same flow as alignment.cpp:191-335: load PLY clouds (:212,241), d_dist = tau_d * max bbox
 * extent of the full model (:246-253), voxel-grid every scene at scene_leaf_size and every
 * model at its d_dist (:265-288), ppf_registration (:290-298), and with --validation_files
 * compare each result with the ground-truth 4x4 (one text file per scene x model) and print
 * 1 or 0 per pair to stdout (:300-335).  Everything else goes to stderr.
 */
#include <getopt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oslam.h"

#define MAX_FILES 256

static int split(char *arg, char **out, int max)
{
    int n = 0;
    char *tok = strtok(arg, ",");
    while (tok && n < max) { if (*tok) out[n++] = tok; tok = strtok(NULL, ","); }
    return n;
}

static int parse_bool(const char *s)
{
    return !strcmp(s, "1") || !strcmp(s, "true") || !strcmp(s, "yes") || !strcmp(s, "on");
}

typedef struct { float *xyz, *nrm; size_t n; } cloud;

static int downsample(cloud *c, float leaf, int dev)
{
    float *x = (float *)malloc(sizeof(float) * 3 * c->n), *q = (float *)malloc(sizeof(float) * 3 * c->n);
    size_t k = 0;
    int rc = oslam_voxel_grid(c->xyz, c->nrm, c->n, 12, leaf, dev, x, q, c->n, &k);
    if (rc != OSLAM_OK) { free(x); free(q); return rc; }
    oslam_free(c->xyz);
    oslam_free(c->nrm);
    c->xyz = x; c->nrm = q; c->n = k;
    return OSLAM_OK;
}

int main(int argc, char **argv)
{
    static struct option opts[] = {
        {"dev", 1, 0, 'd'}, {"logfile", 1, 0, 'L'}, {"loglevel", 1, 0, 'l'}, {"tau_d", 1, 0, 't'},
        {"scene_leaf_size", 1, 0, 's'}, {"ref_point_df", 1, 0, 'r'}, {"vote_count_threshold", 1, 0, 'v'},
        {"cpu_clustering", 1, 0, 'c'}, {"use_l1_norm", 1, 0, '1'}, {"use_averaged_clusters", 1, 0, 'a'},
        {"validation_translation_threshold", 1, 0, 'T'}, {"validation_rotation_threshold", 1, 0, 'R'},
        {"scene_files", 1, 0, 'S'}, {"model_files", 1, 0, 'M'}, {"validation_files", 1, 0, 'V'},
        {"show_normals", 1, 0, 'n'}, {"visualize", 1, 0, 'z'}, {"help", 0, 0, 'h'}, {0, 0, 0, 0}};
    char *scene_files[MAX_FILES], *model_files[MAX_FILES], *val_files[MAX_FILES], *tau_strs[MAX_FILES];
    int n_scenes = 0, n_models = 0, n_val = 0, n_tau = 0, dev = 1, cpu_clustering = 0, l1 = 0, averaged = 0, o, i, j;
    float scene_leaf = 10.0f, thresh = 0.4f, vt = 0.1f, vr = 12.0f;
    unsigned df = 1;
    cloud scenes[MAX_FILES], models[MAX_FILES];
    float d_dists[MAX_FILES], tau[MAX_FILES], *T;
    const float *sx[MAX_FILES], *sn[MAX_FILES], *mx[MAX_FILES], *mn[MAX_FILES];
    size_t sc[MAX_FILES], mc[MAX_FILES];
    int rc;

    while ((o = getopt_long(argc, argv, "", opts, NULL)) != -1) {
        switch (o) {
        case 'd': dev = atoi(optarg); break;
        case 't': n_tau = split(optarg, tau_strs, MAX_FILES); break;
        case 's': scene_leaf = (float)atof(optarg); break;
        case 'r': df = (unsigned)strtoul(optarg, NULL, 10); break;
        case 'v': thresh = (float)atof(optarg); break;
        case 'c': cpu_clustering = parse_bool(optarg); break;
        case '1': l1 = parse_bool(optarg); break;
        case 'a': averaged = parse_bool(optarg); break;
        case 'T': vt = (float)atof(optarg); break;
        case 'R': vr = (float)atof(optarg); break;
        case 'S': n_scenes = split(optarg, scene_files, MAX_FILES); break;
        case 'M': n_models = split(optarg, model_files, MAX_FILES); break;
        case 'V': n_val = split(optarg, val_files, MAX_FILES); break;
        case 'L': case 'l': case 'n': case 'z': break;          /* logging / display flags: accepted, unused */
        default:
            fprintf(stderr, "usage: %s --scene_files a.ply[,..] --model_files m.ply[,..] --tau_d t[,..] [options]\n", argv[0]);
            return 1;
        }
    }
    if (!n_scenes || !n_models || !n_tau) { fprintf(stderr, "--scene_files, --model_files and --tau_d are required\n"); return 1; }
    if (n_tau != n_models) { fprintf(stderr, "Each model must have an associated tau_d.\n"); return 1; }   /* :232-235 */
    if (n_val && n_val != n_scenes * n_models) { fprintf(stderr, "need one validation file per scene x model\n"); return 1; }

    for (i = 0; i < n_scenes; i++) {
        fprintf(stderr, "Loading scene point cloud: %s\n", scene_files[i]);
        if (oslam_ply_read(scene_files[i], &scenes[i].xyz, &scenes[i].nrm, &scenes[i].n) != OSLAM_OK) {
            fprintf(stderr, "Error loading scene file!\n");
            return 1;
        }
    }
    for (j = 0; j < n_models; j++) {
        tau[j] = (float)atof(tau_strs[j]);
        fprintf(stderr, "Loading model point cloud: %s\n", model_files[j]);
        if (oslam_ply_read(model_files[j], &models[j].xyz, &models[j].nrm, &models[j].n) != OSLAM_OK) {
            fprintf(stderr, "Error loading model file!\n");
            return 1;
        }
        oslam_d_dist_from_cloud(models[j].xyz, models[j].n, 12, tau[j], &d_dists[j]);      /* :246-253 */
        fprintf(stderr, "model %d: d_dist %f\n", j, d_dists[j]);
    }
    fprintf(stderr, "Downsampling...\n");
    for (i = 0; i < n_scenes; i++) {
        size_t before = scenes[i].n;
        if ((rc = downsample(&scenes[i], scene_leaf, dev)) != OSLAM_OK) { fprintf(stderr, "voxel grid: %s\n", oslam_last_error()); return 1; }
        fprintf(stderr, "Scene size before/after filtering: %zu / %zu\n", before, scenes[i].n);
    }
    for (j = 0; j < n_models; j++) {
        size_t before = models[j].n;
        if ((rc = downsample(&models[j], d_dists[j], dev)) != OSLAM_OK) { fprintf(stderr, "voxel grid: %s\n", oslam_last_error()); return 1; }
        fprintf(stderr, "Model size before/after filtering: %zu / %zu\n", before, models[j].n);
    }
    for (i = 0; i < n_scenes; i++) { sx[i] = scenes[i].xyz; sn[i] = scenes[i].nrm; sc[i] = scenes[i].n; }
    for (j = 0; j < n_models; j++) { mx[j] = models[j].xyz; mn[j] = models[j].nrm; mc[j] = models[j].n; }
    T = (float *)calloc((size_t)16 * n_scenes * n_models, sizeof(float));
    rc = oslam_ppf_registration(sx, sn, sc, (size_t)n_scenes, mx, mn, mc, (size_t)n_models, 12, d_dists, df, thresh,
                                cpu_clustering, l1, averaged, dev, NULL, T);
    if (rc != OSLAM_OK) { fprintf(stderr, "ppf_registration: %s\n", oslam_last_error()); return 1; }

    for (i = 0; i < n_scenes; i++)
        for (j = 0; j < n_models; j++) {
            const float *R = T + 16 * (i * n_models + j);
            /* the two lines analyze_mian.py:19-41 parses (two-token prefix, as Boost.Log's) */
            fprintf(stderr, "[oslam] [info] Transformations for %s in %s:\n", model_files[j], scene_files[i]);
            for (o = 0; o < 4; o++) fprintf(stderr, "%10.6f %10.6f %10.6f %10.6f\n", R[4 * o], R[4 * o + 1], R[4 * o + 2], R[4 * o + 3]);
            if (n_val) {                                              /* alignment.cpp:300-335 */
                float truth[16], dist[2];
                FILE *f = fopen(val_files[i * n_models + j], "r");
                int k, ok = f != NULL;
                for (k = 0; ok && k < 16; k++) ok = fscanf(f, "%f", &truth[k]) == 1;
                if (f) fclose(f);
                if (!ok) { fprintf(stderr, "cannot read validation file %s\n", val_files[i * n_models + j]); return 1; }
                oslam_ht_dist(R, truth, dist);
                {
                    const float model_diam = d_dists[j] / tau[j];
                    const float trans_thresh = vt * model_diam, rot_thresh = vr * (float)(M_PI / 180.0);
                    const int match = dist[0] < trans_thresh && dist[1] < rot_thresh;
                    fprintf(stderr, "[oslam] [info] Distance (trans, rot): %f, %f\n", dist[0], dist[1]);
                    printf("%d\n", match);
                }
            }
        }
    return 0;
}
