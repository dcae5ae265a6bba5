/*
 * oracle_matlab.c -- TEST INFRASTRUCTURE: a double-precision restatement of the reference's MATLAB
 * prototype of the Drost path (matlab/drost.m and the functions it calls), function by function.
 * Only tests/ may use it; the product never links or loads it.
 *
 * The prototype cannot run here (no MATLAB/Octave, its key needs a JVM: SURVEY.md 8c), and the
 * reference holds no output of it, so this file is pinned by construction only ("parity unpinned"):
 * it follows the formulas of
 *     matlab/point_pair_feature.m:1-11      -> feature
 *     matlab/my_discretize.m:1-9            -> disc
 *     matlab/model_description.m:17-66      -> the map inside orm_voting_scheme; :6-15 the d_dist rule = orm_d_dist
 *     matlab/trans_model_scene.m:1-41       -> alpha_of (roty/rotz/trans of utils/pk-matlabfns)
 *     matlab/voting_scheme.m:9-94           -> orm_voting_scheme (accumulator, per-reference argmax,
 *                                              selection > 0.9 * max)
 * with one substitution the survey allows: the dictionary key -- SHA-1 of MATLAB's serialisation of
 * the four discretised doubles, first 8 bytes read as a double (model_description.m:54-55) -- is
 * replaced by the four bin indices themselves.  Equal discretised features give equal keys in both;
 * what is lost are SHA-1 collisions and the ~2^-11 of keys that decode to NaN and are skipped
 * (model_description.m:57).
 *
 * What differs from the CUDA path by design of the prototype, and is reproduced here:
 *   - double precision throughout;
 *   - alpha_ind = min(round(alpha_disc / d_angle) + 1, n_angle): alpha + pi == 2 pi falls into the
 *     last bin (voting_scheme.m:74); the CUDA path keeps a 31st bin (kernel.cu:341);
 *   - peaks: per reference point the first maximum in column-major order of its [model point x alpha]
 *     slice (max over rows, then max over columns: voting_scheme.m:83-88), then the reference points
 *     whose maximum exceeds 0.9 of the largest (voting_scheme.m:92-94); the CUDA path thresholds all
 *     cells at 0.4 of the global maximum (model.cu:164-170).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } d3;

static d3 sub3(d3 a, d3 b) { d3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static double nrm3(d3 a) { return sqrt(dot3(a, a)); }

/* MATLAB mod(x, y) for y > 0: x - floor(x / y) * y (result has the sign of y) */
static double mmod(double x, double y) { return x - floor(x / y) * y; }

/* point_pair_feature.m:3-9; real() of a complex acos = the angle with the argument clamped */
static void feature(d3 m1, d3 n1, d3 m2, d3 n2, double F[4])
{
    d3 d = sub3(m2, m1);
    double a2 = dot3(n1, d) / (nrm3(n1) * nrm3(d)), a3 = dot3(n2, d) / (nrm3(n2) * nrm3(d)),
           a4 = dot3(n1, n2) / (nrm3(n1) * nrm3(n2));
    F[0] = nrm3(d);
    F[1] = acos(a2 > 1 ? 1 : a2 < -1 ? -1 : a2);       /* real(acos(x)) for |x| > 1: 0 or pi */
    F[2] = acos(a3 > 1 ? 1 : a3 < -1 ? -1 : a3);
    F[3] = acos(a4 > 1 ? 1 : a4 < -1 ? -1 : a4);
}

/* my_discretize.m:3-4 -> the bin indices of F - mod(F, step); near[i] = -1 / +1 when component i lies within
 * eps bins of its lower / upper bin boundary (single precision may land in that neighbour), else 0 */
static int disc(const double F[4], double d_dist, double d_angle, double eps, int32_t k[4], int near[4])
{
    int i;
    for (i = 0; i < 4; i++) {
        const double step = i == 0 ? d_dist : d_angle, q = F[i] / step, fl = floor(q);
        if (!(q == q) || fabs(q) > 1e9) return 0;          /* NaN: the pair has no key (coincident points) */
        k[i] = (int32_t)fl;
        near[i] = q - fl < eps ? -1 : (fl + 1.0 - q < eps ? 1 : 0);
    }
    return 1;
}

/* d_dist rule of model_description.m:6-15: a tenth of the largest distance from the bounding box's centre */
double orm_d_dist(const double *pts, int M)
{
    double lo[3], hi[3], c[3], best = 0;
    int i, a;
    for (a = 0; a < 3; a++) lo[a] = hi[a] = pts[a];
    for (i = 1; i < M; i++)
        for (a = 0; a < 3; a++) {
            if (pts[3 * i + a] < lo[a]) lo[a] = pts[3 * i + a];
            if (pts[3 * i + a] > hi[a]) hi[a] = pts[3 * i + a];
        }
    for (a = 0; a < 3; a++) c[a] = 0.5 * (lo[a] + hi[a]);
    for (i = 0; i < M; i++) {
        double dx = pts[3 * i] - c[0], dy = pts[3 * i + 1] - c[1], dz = pts[3 * i + 2] - c[2];
        double dd = sqrt(dx * dx + dy * dy + dz * dz);
        if (dd > best) best = dd;
    }
    return 0.1 * best;
}

/* roty / rotz / trans of utils/pk-matlabfns applied as trans_model_scene.m:13-17: T_g = rotz * roty * trans,
 * returned as the three rows acting on (p - r) */
static void frame(d3 n, double R[9])
{
    const double ty = atan2(n.z, n.x), cy = cos(ty), sy = sin(ty);
    /* n_tmp = roty(ty) * n */
    const double tx_ = cy * n.x + sy * n.z, ty_ = n.y;
    const double tz = -atan2(ty_, tx_), cz = cos(tz), sz = sin(tz);
    /* roty = [c 0 s; 0 1 0; -s 0 c], rotz = [c -s 0; s c 0; 0 0 1]; R = rotz * roty */
    R[0] = cz * cy;  R[1] = -sz; R[2] = cz * sy;
    R[3] = sz * cy;  R[4] = cz;  R[5] = sz * sy;
    R[6] = -sy;      R[7] = 0;   R[8] = cy;
}

/* trans_model_scene.m:29-39: alpha = atan2(w' * cross(u_hat, v_hat), u_hat' * v_hat), w = x axis */
static double alpha_of(d3 m_r, d3 n_m, d3 m_i, d3 s_r, d3 n_s, d3 s_i)
{
    double Rm[9], Rs[9];
    d3 a = sub3(m_i, m_r), b = sub3(s_i, s_r);
    double uy, uz, vy, vz;
    frame(n_m, Rm);
    frame(n_s, Rs);
    uy = Rm[3] * a.x + Rm[4] * a.y + Rm[5] * a.z;
    uz = Rm[6] * a.x + Rm[7] * a.y + Rm[8] * a.z;
    vy = Rs[3] * b.x + Rs[4] * b.y + Rs[5] * b.z;
    vz = Rs[6] * b.x + Rs[7] * b.y + Rs[8] * b.z;
    return atan2(uy * vz - uz * vy, uy * vy + uz * vz);
}

/* a model pair under its key; alt: listed under a neighbouring key because it lies within eps of that boundary
 * (not part of the prototype's map: only counts towards the tolerance); edge: its own key is that uncertain */
typedef struct { int32_t k[4]; int32_t mr, mi; int alt, edge; } mpair;

static int mpair_order(const void *a, const void *b)
{
    const mpair *x = (const mpair *)a, *y = (const mpair *)b;
    int i;
    for (i = 0; i < 4; i++)
        if (x->k[i] != y->k[i]) return x->k[i] < y->k[i] ? -1 : 1;
    if (x->mr != y->mr) return x->mr < y->mr ? -1 : 1;
    return x->mi < y->mi ? -1 : (x->mi > y->mi);
}

static int key_cmp(const int32_t a[4], const int32_t b[4])
{
    int i;
    for (i = 0; i < 4; i++)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

/*
 * voting_scheme.m with model_description.m's map built inside.
 *   pts/nrm: double [n][3].  skip: voting_scheme.m:11 (5 in the prototype).  eps_bins: a discretised quantity
 *   within eps_bins of a bin boundary is a "bin-edge case": single precision may legitimately land in the
 *   neighbouring bin there.
 * Outputs (caller-allocated):
 *   acc        [n_ref][M][30] uint32: accumulator(model point, alpha_ind, reference point) (voting_scheme.m:76-77),
 *              n_ref = ceil(S / skip), reference point t = scene point t * skip
 *   argmax_row [n_ref], argmax_col [n_ref], max_tots [n_ref]: I_rows, I_cols, max_tots (voting_scheme.m:83-88),
 *              0-based, -1 for a reference point without votes
 *   selected   [n_ref] uint8: max_tots / max(max_tots) > 0.9 (voting_scheme.m:90-92)
 *   edge_votes [n_ref] uint64: a bound on what bin-edge cases can change in the reference point's slice, as a
 *              number of votes: votes cast here whose key (model or scene side) or alpha is a bin-edge case (each
 *              may move or vanish) plus the votes a bin-edge pair WOULD cast under its neighbouring key (each
 *              may appear).  The accumulators of a single-precision run differ from acc by at most
 *              2 * edge_votes in the L1 norm.
 * Returns the total number of votes, or (uint64_t)-1 without memory.
 */
uint64_t orm_voting_scheme(const double *m_pts, const double *m_nrm, int M, const double *s_pts, const double *s_nrm, int S,
                           int skip, double d_dist, double eps_bins, uint32_t *acc, int32_t *argmax_row, int32_t *argmax_col,
                           uint32_t *max_tots, uint8_t *selected, uint64_t *edge_votes)
{
    const int n_angle = 30;
    const double d_angle = 2.0 * M_PI / n_angle;
    const int n_ref = (S + skip - 1) / skip;
    size_t n_pairs = 0, p;
    uint64_t total = 0;
    uint32_t best = 0;
    int i, j, t, c;
    /* every pair once under its key + once per near boundary under the neighbouring key */
    mpair *mp = (mpair *)malloc(sizeof(mpair) * 5 * (size_t)M * (size_t)(M > 1 ? M - 1 : 1));
    if (!mp) return (uint64_t)-1;
    /* model_description.m:28-66: every ordered pair i != j under the key of its discretised feature */
    for (i = 0; i < M; i++)
        for (j = 0; j < M; j++) {
            double F[4];
            int32_t k[4];
            int near[4], any;
            d3 a = {m_pts[3 * i], m_pts[3 * i + 1], m_pts[3 * i + 2]}, na = {m_nrm[3 * i], m_nrm[3 * i + 1], m_nrm[3 * i + 2]};
            d3 b = {m_pts[3 * j], m_pts[3 * j + 1], m_pts[3 * j + 2]}, nb = {m_nrm[3 * j], m_nrm[3 * j + 1], m_nrm[3 * j + 2]};
            if (i == j) continue;                                          /* :36-40 */
            feature(a, na, b, nb, F);
            if (!disc(F, d_dist, d_angle, eps_bins, k, near)) continue;
            any = near[0] || near[1] || near[2] || near[3];
            memcpy(mp[n_pairs].k, k, sizeof k);
            mp[n_pairs].mr = i;
            mp[n_pairs].mi = j;
            mp[n_pairs].alt = 0;
            mp[n_pairs].edge = any;
            n_pairs++;
            for (c = 0; c < 4; c++)
                if (near[c]) {
                    memcpy(mp[n_pairs].k, k, sizeof k);
                    mp[n_pairs].k[c] += near[c];
                    mp[n_pairs].mr = i;
                    mp[n_pairs].mi = j;
                    mp[n_pairs].alt = 1;
                    mp[n_pairs].edge = 1;
                    n_pairs++;
                }
        }
    qsort(mp, n_pairs, sizeof(mpair), mpair_order);
    memset(acc, 0, sizeof(uint32_t) * (size_t)n_ref * (size_t)M * n_angle);
    for (t = 0; t < n_ref; t++) {
        const int r = t * skip;                                            /* r_indices = 1:skip:end, :14 */
        uint32_t *A = acc + (size_t)t * (size_t)M * n_angle;
        d3 s_r = {s_pts[3 * r], s_pts[3 * r + 1], s_pts[3 * r + 2]}, n_r = {s_nrm[3 * r], s_nrm[3 * r + 1], s_nrm[3 * r + 2]};
        uint64_t ev = 0;
        uint32_t mx = 0;
        int brow = -1, bcol = -1, col, row;
        for (i = 0; i < S; i++) {
            double F[4];
            int32_t k[4], k2[4];
            int near[4], sany;
            size_t lo, hi;
            d3 s_i = {s_pts[3 * i], s_pts[3 * i + 1], s_pts[3 * i + 2]}, n_i = {s_nrm[3 * i], s_nrm[3 * i + 1], s_nrm[3 * i + 2]};
            if (i == r) continue;                                          /* :38-40 */
            feature(s_r, n_r, s_i, n_i, F);
            if (!disc(F, d_dist, d_angle, eps_bins, k, near)) continue;
            sany = near[0] || near[1] || near[2] || near[3];
            /* what the pair would match under a neighbouring key: may appear in a single-precision run */
            for (c = 0; c < 4; c++)
                if (near[c]) {
                    memcpy(k2, k, sizeof k);
                    k2[c] += near[c];
                    for (lo = 0, hi = n_pairs; lo < hi;) {
                        size_t mid = lo + (hi - lo) / 2;
                        if (key_cmp(mp[mid].k, k2) < 0) lo = mid + 1; else hi = mid;
                    }
                    for (p = lo; p < n_pairs && key_cmp(mp[p].k, k2) == 0; p++) ev++;
                }
            for (lo = 0, hi = n_pairs; lo < hi;) {                         /* isKey(model_map, key), :53 */
                size_t mid = lo + (hi - lo) / 2;
                if (key_cmp(mp[mid].k, k) < 0) lo = mid + 1; else hi = mid;
            }
            for (p = lo; p < n_pairs && key_cmp(mp[p].k, k) == 0; p++) {   /* :58-81 */
                d3 m_r, n_m, m_i;
                double alpha, ap, alpha_disc, fr;
                int alpha_ind;
                if (mp[p].alt) { ev++; continue; }                         /* not in the prototype's map: tolerance only */
                m_r.x = m_pts[3 * mp[p].mr]; m_r.y = m_pts[3 * mp[p].mr + 1]; m_r.z = m_pts[3 * mp[p].mr + 2];
                n_m.x = m_nrm[3 * mp[p].mr]; n_m.y = m_nrm[3 * mp[p].mr + 1]; n_m.z = m_nrm[3 * mp[p].mr + 2];
                m_i.x = m_pts[3 * mp[p].mi]; m_i.y = m_pts[3 * mp[p].mi + 1]; m_i.z = m_pts[3 * mp[p].mi + 2];
                alpha = alpha_of(m_r, n_m, m_i, s_r, n_r, s_i);
                ap = alpha + M_PI;
                alpha_disc = ap - mmod(ap, d_angle);                                            /* :72 */
                alpha_ind = (int)floor(alpha_disc / d_angle + 0.5) + 1;                         /* round(), :74 */
                fr = ap / d_angle - floor(ap / d_angle);
                if (alpha_ind > n_angle) alpha_ind = n_angle;
                if (alpha_ind < 1) alpha_ind = 1;
                A[(size_t)mp[p].mr * n_angle + (alpha_ind - 1)]++;
                total++;
                if (fr < eps_bins || 1.0 - fr < eps_bins || sany || mp[p].edge) ev++;
            }
        }
        /* [Y_rows, I_row] = max(accumulator(:,:,r)); [max_tot, I_col] = max(Y_rows), :83-88: per column the first
         * row with the column's maximum, then the first column with the largest of them */
        for (col = 0; col < n_angle; col++) {
            uint32_t cm = 0;
            int cr = 0;
            for (row = 0; row < M; row++)
                if (A[(size_t)row * n_angle + col] > cm) { cm = A[(size_t)row * n_angle + col]; cr = row; }
            if (cm > mx) { mx = cm; brow = cr; bcol = col; }
        }
        argmax_row[t] = brow;
        argmax_col[t] = bcol;
        max_tots[t] = mx;
        if (edge_votes) edge_votes[t] = ev;
        if (mx > best) best = mx;
    }
    for (t = 0; t < n_ref; t++) selected[t] = best > 0 && (double)max_tots[t] / (double)best > 0.9;   /* :90-92 */
    free(mp);
    return total;
}
