/*
 * oracle/csrc/objchain.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The object's gradient chain of one optimisation step, written out operation
 * by operation with ORDER-INDEPENDENT sums, so that the CPU oracle's free-running
 * trajectory of the object pose (reference loop homan/jointopt.py:158-192) is a
 * defined function of the inputs: the same on 8 and on 64 host threads, and the
 * same bits as the HIP kernels produce (csrc/raster.hip k_bwd_sweep,
 * csrc/geometry.hip k_rigid_bwd<true>, csrc/adam.hip).
 *
 *   silhouette loss -> per-sample gradient -> NMR edge-sweep pseudo-gradient per
 *   (face, corner) [orc_nmr_grad_faces_alpha_exact] -> vertex gather ->
 *   projection backward -> + the other per-vertex terms -> per-frame sums of the
 *   rigid backward -> rot6d backward [orc_rigid_bwd_sil_exact].
 *
 * Same mathematics as the faithful restatement (orc_nmr_grad_faces_alpha in
 * nmr_raster.c + torch autograd through oracle/model.py, pinned against the
 * reference's goldens); tests/test_objchain.py holds the two within fp32
 * rounding of each other.  What differs is only what fp32 leaves open:
 *   - every reduction rounds its addends to multiples of q = 2^log2q
 *     ((x + M) - M in double with M = 1.5 * 2^(52 + log2q)) and adds them in
 *     double: exact while |sum| < 2^53 q, hence independent of the order;
 *   - per-term arithmetic in one fixed order of IEEE operations:
 *       c = num / (p1 - d0);  k = (c * 2) / is;  dist = k * (d1 - cross) +- eps;
 *       term = diff / dist
 *     (the faithful loop evaluates num / (p1 - d0) * (d1 - cross) * 2 / is).
 * Build with -ffp-contract=off (the Makefile does).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float oc_topix(float v, int is)
{
    float a = v * (float)is;
    a = a + (float)is;
    a = a - 1.0f;
    return 0.5f * a;
}
static inline int oc_backside(const float *f)
{
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}
static inline int oc_imax(int a, int b) { return a > b ? a : b; }
static inline int oc_imin(int a, int b) { return a < b ? a : b; }

double orc_sum_magic(int log2q)
{
    return 6755399441055744.0 /* 1.5 * 2^52 */ * ldexp(1.0, log2q ? log2q : -44);
}
static inline double oc_quant(float x, double magic)
{
    volatile double y = (double)x + magic;      /* (volatile: the sum is rounded to double before the subtraction) */
    return y - magic;
}

/*
 * faces (B, 2F, 9): the F mesh faces, then their reversed-winding copies (vertex
 * order [2,1,0]: nr fill_back).  idx_map (B,is,is) of orc_nmr_face_index_map on
 * those 2F faces.  grad_alpha (B,is,is): dL/dalpha per SAMPLE (unflipped grid).
 * parts (B, F, 3 mesh corners, 2): d loss / d NDC (x, y), both windings of a mesh
 * face accumulated onto its mesh corners (corner k of the reversed copy is mesh
 * corner 2 - k), as exact sums on the grid 2^log2q.
 */
void orc_nmr_grad_faces_alpha_exact(const float *faces, const int32_t *idx_map, const float *grad_alpha, int B, int F,
                                    int is, float eps, int log2q, double *parts)
{
    const long npix = (long)is * is;
    const double magic = orc_sum_magic(log2q);
    const int NF = 2 * F;
    memset(parts, 0, sizeof(double) * (size_t)B * F * 6);
#pragma omp parallel for schedule(dynamic, 16)
    for (long bfi = 0; bfi < (long)B * F; ++bfi) {
        const int bn = (int)(bfi / F), fi = (int)(bfi % F);
        double *out = parts + bfi * 6;
        const int32_t *idx = idx_map + bn * npix;
        const float *ga = grad_alpha + bn * npix;
        for (int var = 0; var < 2; ++var) {
            const int fn = fi + var * F;
            const float *f = faces + ((long)bn * NF + fn) * 9;
            if (oc_backside(f)) continue;
            for (int e = 0; e < 3; ++e) {
                const int pi[3] = {e, (e + 1) % 3, (e + 2) % 3};
                float pp[3][2];
                for (int n = 0; n < 3; ++n)
                    for (int d = 0; d < 2; ++d) pp[n][d] = oc_topix(f[3 * pi[n] + d], is);
                /* mesh corners the two end points of this edge belong to */
                const int m0 = var ? 2 - pi[0] : pi[0], m1 = var ? 2 - pi[1] : pi[1];
                for (int axis = 0; axis < 2; ++axis) {
                    float p[3][2];
                    for (int n = 0; n < 3; ++n) { p[n][0] = pp[n][axis]; p[n][1] = pp[n][1 - axis]; }
                    if (p[0][0] == p[1][0]) continue;
                    int dir;
                    if (axis == 0) dir = (p[0][0] < p[1][0]) ? -1 : 1;
                    else           dir = (p[0][0] < p[1][0]) ? 1 : -1;
                    const int step = (axis == 0) ? is : 1;
                    const int d0_from = (int)fmaxf(ceilf(fminf(p[0][0], p[1][0])), 0.0f);
                    const int d0_to = (int)fminf(fmaxf(p[0][0], p[1][0]), (float)is - 1.0f);
                    const int comp = 1 - axis;          /* row sweeps (axis 1) move x, column sweeps (axis 0) move y */
                    double *acc0 = out + 2 * m0 + comp, *acc1 = out + 2 * m1 + comp;
                    const float slope = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
                    const float num = p[1][0] - p[0][0];
                    for (int d0 = d0_from; d0 <= d0_to; ++d0) {
                        const float d1_cross = slope * ((float)d0 - p[0][0]) + p[0][1];
                        if (!(d1_cross > -8.0f && d1_cross < (float)is + 8.0f)) continue;
                        const int d1_in = (dir > 0) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
                        const int d1_out = d1_in + dir;
                        if (d1_in < 0 || is <= d1_in) continue;
                        if (d1_out < 0 || is <= d1_out) continue;
                        const long base = (axis == 0) ? (long)d0 : (long)d0 * is;
                        const int alpha_out_set = idx[base + (long)d1_out * step] >= 0;
                        const int in_is_fn = idx[base + (long)d1_in * step] == fn;
                        const int use0 = p[1][0] != (float)d0;
                        const int use1 = p[0][0] != (float)d0;
                        const float c0 = use0 ? num / (p[1][0] - (float)d0) : 0.0f;
                        const float c1 = use1 ? num / ((float)d0 - p[0][0]) : 0.0f;
                        const float k0 = (c0 * 2.0f) / (float)is;
                        const float k1 = (c1 * 2.0f) / (float)is;
                        /* ---- outward sweep */
                        if (in_is_fn) {
                            const int lim = (dir > 0) ? is - 1 : 0;
                            const int from = oc_imax(oc_imin(d1_out, lim), 0);
                            const int to = oc_imin(oc_imax(d1_out, lim), is - 1);
                            for (int d1 = from; d1 <= to; ++d1) {
                                const long q = base + (long)d1 * step;
                                if (idx[q] >= 0) continue;                 /* (a - alpha_in) = 0 */
                                const float diff = -ga[q];
                                if (!(diff > 0.0f)) continue;
                                const float t = (float)d1 - d1_cross;
                                float dist0 = k0 * t, dist1 = k1 * t;
                                dist0 += (0.0f < dist0) ? eps : -eps;
                                dist1 += (0.0f < dist1) ? eps : -eps;
                                if (use0) *acc0 -= oc_quant(diff / dist0, magic);
                                if (use1) *acc1 -= oc_quant(diff / dist1, magic);
                            }
                        }
                        /* ---- inward sweep */
                        if (!alpha_out_set) {
                            float c2;
                            if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0.0f)
                                c2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                            else
                                c2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                            if (!(c2 == c2)) continue;
                            c2 = fminf(fmaxf(c2, -4.0f), (float)is + 4.0f);
                            const int lim = (dir > 0) ? (int)ceilf(c2) : (int)floorf(c2);
                            const int from = oc_imax(oc_imin(d1_in, lim), 0);
                            const int to = oc_imin(oc_imax(d1_in, lim), is - 1);
                            for (int d1 = from; d1 <= to; ++d1) {
                                const long q = base + (long)d1 * step;
                                if (idx[q] != fn) continue;
                                const float diff = ga[q];
                                if (!(diff > 0.0f)) continue;
                                const float t = (float)d1 - d1_cross;
                                float dist0 = k0 * t, dist1 = k1 * t;
                                dist0 += (0.0f < dist0) ? eps : -eps;
                                dist1 += (0.0f < dist1) ? eps : -eps;
                                if (use0) *acc0 -= oc_quant(diff / dist0, magic);
                                if (use1) *acc1 -= oc_quant(diff / dist1, magic);
                            }
                        }
                    }
                }
            }
        }
    }
}

/* rot6d (3x2 row-major) -> R (3x3 row-major), reference homan/utils/geometry.py:9-27, order of oracle/model.py rot6d_to_matrix */
void oc_rot6d_to_mat(const float *r6, float *R)
{
    const float a1[3] = {r6[0], r6[2], r6[4]}, a2[3] = {r6[1], r6[3], r6[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float nu = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / nu, u[1] / nu, u[2] / nu};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    for (int i = 0; i < 3; ++i) { R[3 * i] = b1[i]; R[3 * i + 1] = b2[i]; R[3 * i + 2] = b3[i]; }
}

/* dL/dR -> dL/drot6d: the chain rule through the Gram-Schmidt construction above, one fixed order of operations */
void oc_rot6d_backward(const float *r6, const float *dR, float *dr6)
{
    const float a1[3] = {r6[0], r6[2], r6[4]}, a2[3] = {r6[1], r6[3], r6[5]};
    const float n1r = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    const float n1 = fmaxf(n1r, 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float nur = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float nu = fmaxf(nur, 1e-12f);
    const float b2[3] = {u[0] / nu, u[1] / nu, u[2] / nu};
    float db1[3] = {dR[0], dR[3], dR[6]}, db2[3] = {dR[1], dR[4], dR[7]};
    const float db3[3] = {dR[2], dR[5], dR[8]};
    db1[0] += b2[1] * db3[2] - b2[2] * db3[1];
    db1[1] += b2[2] * db3[0] - b2[0] * db3[2];
    db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
    db2[0] += db3[1] * b1[2] - db3[2] * b1[1];
    db2[1] += db3[2] * b1[0] - db3[0] * b1[2];
    db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
    float du[3];
    if (nur > 1e-12f) {
        const float s = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
        for (int i = 0; i < 3; ++i) du[i] = (db2[i] - b2[i] * s) / nu;
    } else {
        for (int i = 0; i < 3; ++i) du[i] = db2[i] / nu;
    }
    float da2[3] = {du[0], du[1], du[2]};
    const float dd = -(du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2]);
    for (int i = 0; i < 3; ++i) { db1[i] += -d * du[i] + dd * a2[i]; da2[i] += dd * b1[i]; }
    float da1[3];
    if (n1r > 1e-12f) {
        const float s = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
        for (int i = 0; i < 3; ++i) da1[i] = (db1[i] - b1[i] * s) / n1;
    } else {
        for (int i = 0; i < 3; ++i) da1[i] = db1[i] / n1;
    }
    for (int i = 0; i < 3; ++i) { dr6[2 * i] = da1[i]; dr6[2 * i + 1] = da2[i]; }
}

/*
 * Backward of verts = (s * mesh) @ R(rot6d) + t (reference homan/utils/camera.py:108-139 through homan/homan.py:298-307)
 * for per-vertex gradients  gf = sum_k w[k] * terms[k]  [+ the silhouette term: per-corner sums `parts` gathered over the
 * vertex's corners (CSR adj_off / adj_items, item = face * 3 + corner) and pushed through the backward of
 * nr.projection (K (B,3,3), orig_size; zero distortion)].
 * mesh (B,V,3), rot6d (B,6), scale: one value (abs taken if abs_scale), terms: n_terms arrays (B,V,3).
 * Outputs g_rot6d (B,6), g_trans (B,3), g_scale_part (B) (optional): the 13 per-frame sums over the vertices are exact
 * sums on the grid 2^log2q.
 */
void orc_rigid_bwd_sil_exact(const float *mesh, const float *rot6d, float scale_raw, int abs_scale, const float *const *terms,
                             const float *weights, int n_terms, const double *parts, const int32_t *adj_off,
                             const int32_t *adj_items, const float *cam_verts, const float *K, float orig_size, int F, int B,
                             int V, int log2q, float *g_rot6d, float *g_trans, float *g_scale_part, float *g_verts_out)
{
    const double magic = orc_sum_magic(log2q);
    const float s = abs_scale ? fabsf(scale_raw) : scale_raw;
    for (int n = 0; n < B; ++n) {
        float R[9];
        oc_rot6d_to_mat(rot6d + n * 6, R);
        double acc[13];
        for (int k = 0; k < 13; ++k) acc[k] = 0.0;
        for (int v = 0; v < V; ++v) {
            const long o = ((long)n * V + v) * 3;
            const float m[3] = {mesh[o], mesh[o + 1], mesh[o + 2]};
            float gf[3] = {0.f, 0.f, 0.f}, gt[3];
            for (int k = 0; k < n_terms; ++k) {
                gf[0] += weights[k] * terms[k][o];
                gf[1] += weights[k] * terms[k][o + 1];
                gf[2] += weights[k] * terms[k][o + 2];
            }
            if (parts) {
                double su = 0.0, sv = 0.0;
                const double *pf = parts + (long)n * F * 6;
                for (int a = adj_off[v]; a < adj_off[v + 1]; ++a) {
                    su += pf[2 * adj_items[a]];
                    sv += pf[2 * adj_items[a] + 1];
                }
                const float gu = (float)su, gv = (float)sv;
                const float *k = K + n * 9;
                const float x = cam_verts[o], y = cam_verts[o + 1], z = cam_verts[o + 2];
                const float zz = z + 1e-9f;
                const float du0 = gu * (2.0f / orig_size), dv0 = -gv * (2.0f / orig_size);
                const float dxn = k[0] * du0 + k[3] * dv0;
                const float dyn = k[1] * du0 + k[4] * dv0;
                gf[0] += dxn / zz;
                gf[1] += dyn / zz;
                gf[2] += -(dxn * x + dyn * y) / (zz * zz);
            }
            gt[0] = gf[0] + 0.f; gt[1] = gf[1] + 0.f; gt[2] = gf[2] + 0.f;
            if (g_verts_out) { g_verts_out[o] = gf[0]; g_verts_out[o + 1] = gf[1]; g_verts_out[o + 2] = gf[2]; }
            const float dm[3] = {R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2], R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2],
                                 R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) acc[3 * i + j] += oc_quant((s * m[i]) * gt[j], magic);
            for (int j = 0; j < 3; ++j) acc[9 + j] += oc_quant(gt[j], magic);
            acc[12] += oc_quant(m[0] * dm[0] + m[1] * dm[1] + m[2] * dm[2], magic);
        }
        float tot[13];
        for (int k = 0; k < 13; ++k) tot[k] = (float)acc[k];
        oc_rot6d_backward(rot6d + n * 6, tot, g_rot6d + n * 6);
        for (int k = 0; k < 3; ++k) g_trans[n * 3 + k] = tot[9 + k];
        if (g_scale_part) g_scale_part[n] = (abs_scale && scale_raw < 0.f) ? -tot[12] : tot[12];
    }
}
