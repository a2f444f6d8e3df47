/*
 * oracle/csrc/nmr_raster.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32, no FMA contraction: build with
 * -ffp-contract=off) of the hard z-buffer rasteriser + edge-sweep
 * pseudo-gradient of the Neural 3D Mesh Renderer (Kato et al., CVPR 2018) as
 * the reference uses it for `mode="silhouettes"`:
 *   reference call sites: homan/losses.py:73-77 (Renderer ctor, defaults
 *   anti_aliasing=True, fill_back=True, near=0.1, far=100, eps=1e-3),
 *   homan/losses.py:187 (silhouette render), homan/homan.py:168-172.
 * The renderer itself is the third-party package `neural_renderer`
 * (hassony2/multiperson fork @ HEAD, un-vendored: reference README.md:54-60),
 * whose source is NOT in /root/reference.  PARITY UNPINNED: this file restates
 * the published algorithm (per-face barycentric inverse, per-pixel inclusive
 * edge test + perspective-correct z-min, per-(face,edge,axis) line sweep
 * backward) with every convention written down below.
 *
 * Conventions fixed here (and mirrored by the HIP product kernels):
 *   - `faces` is (B, NF, 3, 3) fp32: per vertex (x, y) in NDC [-1,1] (y up) and
 *     z = camera depth.  fill_back doubling is done by the caller.
 *   - sample (yi, xi) of an `is` x `is` grid sits at
 *       xp = (2*xi + 1 - is) / is,  yp = (2*yi + 1 - is) / is
 *     which is pixel coordinate p = 0.5*(x*is + is - 1) == xi.
 *   - back-facing iff (y2-y0)*(x1-x0) < (y1-y0)*(x2-x0)  (skipped).
 *   - inside iff none of the three edge functions is strictly negative.
 *   - barycentrics w_k from the 3x3 inverse in pixel coordinates, clamped to
 *     [0,1]; perspective-correct depth z = (sum_k w_k) / (sum_k w_k * (1/z_k)),
 *     i.e. 1/z = sum_k (w_k / sum w) / z_k of the published algorithm with the
 *     renormalisation folded into one division per sample (reciprocal vertex
 *     depths 1/z_k are formed once per face); hit iff near < z < far.
 *   - z-buffer: strictly smaller z wins; ties keep the LOWEST face index
 *     (the sequential per-pixel loop order of the upstream kernel).
 *   - zero-area faces (barycentric denominator == 0) are skipped
 *     (upstream would propagate inf/nan; documented divergence); so are faces
 *     with a vertex projected beyond |1e15| or not finite (same reason).
 *   - edges whose two end points share the sweep coordinate are skipped in the
 *     backward (upstream divides by zero there).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* a vertex projected beyond 1e15 (or not finite): the edge functions would overflow to inf - inf; such a face is culled
 * (the product kernels cull it in their face setup) */
static inline int orc_insane(const float *f)
{
    for (int k = 0; k < 3; ++k)
        if (!(fabsf(f[3 * k]) <= 1e15f && fabsf(f[3 * k + 1]) <= 1e15f)) return 1;
    return 0;
}

static inline int orc_backside(const float *f)
{
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

static inline float orc_topix(float v, int is)
{
    /* 0.5 * (v*is + is - 1), evaluated left to right in fp32 */
    float a = v * (float)is;
    a = a + (float)is;
    a = a - 1.0f;
    return 0.5f * a;
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/*
 * Forward: face-index map (int32, -1 = background) and optional depth map.
 * Restates the two forward kernels of the NMR rasteriser.  The loop nest is
 * per face over its (conservative) pixel bounding box instead of per pixel
 * over all faces; because faces are visited in ascending index order and a
 * pixel is only overwritten by a strictly smaller depth the result is the one
 * of the upstream per-pixel loop.
 */
void orc_nmr_face_index_map(const float *faces, int B, int NF, int is,
                            float near, float far,
                            int32_t *idx_map, float *depth_map /* B*is*is */)
{
    const long npix = (long)is * is;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        int32_t *idx = idx_map + b * npix;
        float *dep = depth_map + b * npix;
        for (long i = 0; i < npix; ++i) { idx[i] = -1; dep[i] = far; }
        for (int fn = 0; fn < NF; ++fn) {
            const float *f = faces + ((long)b * NF + fn) * 9;
            if (orc_backside(f) || orc_insane(f)) continue;
            float p[3][2];
            for (int k = 0; k < 3; ++k)
                for (int d = 0; d < 2; ++d) p[k][d] = orc_topix(f[3 * k + d], is);
            float inv[9] = {
                p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
            float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                        p[1][0] * (p[2][1] - p[0][1]);
            if (den == 0.0f) continue;
            for (int k = 0; k < 9; ++k) inv[k] /= den;
            const float rz0 = 1.0f / f[2], rz1 = 1.0f / f[5], rz2 = 1.0f / f[8];
            float xmin = fminf(p[0][0], fminf(p[1][0], p[2][0]));
            float xmax = fmaxf(p[0][0], fmaxf(p[1][0], p[2][0]));
            float ymin = fminf(p[0][1], fminf(p[1][1], p[2][1]));
            float ymax = fmaxf(p[0][1], fmaxf(p[1][1], p[2][1]));
            if (!(xmax >= -2.0f && ymax >= -2.0f && xmin <= is + 1.0f && ymin <= is + 1.0f))
                continue; /* also rejects NaN */
            int x0 = imax(0, (int)floorf(fmaxf(xmin, -2.0f)) - 1);
            int x1 = imin(is - 1, (int)ceilf(fminf(xmax, is + 1.0f)) + 1);
            int y0 = imax(0, (int)floorf(fmaxf(ymin, -2.0f)) - 1);
            int y1 = imin(is - 1, (int)ceilf(fminf(ymax, is + 1.0f)) + 1);
            for (int yi = y0; yi <= y1; ++yi) {
                const float yp = (float)(2 * yi + 1 - is) / (float)is;
                for (int xi = x0; xi <= x1; ++xi) {
                    const float xp = (float)(2 * xi + 1 - is) / (float)is;
                    if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                        ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                        ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                        continue;
                    float w[3], ws = 0.0f;
                    for (int k = 0; k < 3; ++k) {
                        float t = inv[3 * k + 0] * (float)xi;
                        t = t + inv[3 * k + 1] * (float)yi;
                        t = t + inv[3 * k + 2];
                        t = fminf(fmaxf(t, 0.0f), 1.0f);
                        w[k] = t;
                        ws += t;
                    }
                    float s = w[0] * rz0;
                    s = s + w[1] * rz1;
                    s = s + w[2] * rz2;
                    const float zp = ws / s;
                    if (!(zp > near && zp < far)) continue;
                    const long pix = (long)yi * is + xi;
                    if (zp < dep[pix]) { dep[pix] = zp; idx[pix] = fn; }
                }
            }
        }
    }
}

/*
 * Backward of alpha w.r.t. the NDC (x, y) of the face vertices: the NMR
 * edge-sweep pseudo-gradient.  alpha is 1 where idx_map >= 0, else 0.
 * grad_alpha is dL/dalpha on the same `is` grid.  grad_faces (B, NF, 9) is
 * overwritten (z components stay 0).
 */
void orc_nmr_grad_faces_alpha(const float *faces, const int32_t *idx_map,
                              const float *grad_alpha, int B, int NF, int is,
                              float eps, float *grad_faces)
{
    const long npix = (long)is * is;
#pragma omp parallel for schedule(dynamic, 16)
    for (long bf = 0; bf < (long)B * NF; ++bf) {
        const int bn = (int)(bf / NF);
        const int fn = (int)(bf % NF);
        const float *f = faces + bf * 9;
        const int32_t *idx = idx_map + bn * npix;
        const float *ga = grad_alpha + bn * npix;
        float g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float *out = grad_faces + bf * 9;
        if (orc_backside(f) || orc_insane(f)) { memset(out, 0, 9 * sizeof(float)); continue; }   /* (culled in every pass, like the kernels' face setup) */

        for (int e = 0; e < 3; ++e) {
            const int pi[3] = {e, (e + 1) % 3, (e + 2) % 3};
            float pp[3][2];
            for (int n = 0; n < 3; ++n)
                for (int d = 0; d < 2; ++d) pp[n][d] = orc_topix(f[3 * pi[n] + d], is);
            for (int axis = 0; axis < 2; ++axis) {
                /* p[n][0]: sweep coordinate (x for axis 0, y for axis 1); p[n][1]: the other */
                float p[3][2];
                for (int n = 0; n < 3; ++n) { p[n][0] = pp[n][axis]; p[n][1] = pp[n][1 - axis]; }
                if (p[0][0] == p[1][0]) continue; /* degenerate in this axis */
                int dir;
                if (axis == 0) dir = (p[0][0] < p[1][0]) ? -1 : 1;
                else           dir = (p[0][0] < p[1][0]) ? 1 : -1;
                const int step = (axis == 0) ? is : 1;   /* memory stride of d1 */
                const int d0_from = (int)fmaxf(ceilf(fminf(p[0][0], p[1][0])), 0.0f);
                const int d0_to = (int)fminf(fmaxf(p[0][0], p[1][0]), (float)is - 1.0f);
                const int gi0 = pi[0] * 3 + (1 - axis);
                const int gi1 = pi[1] * 3 + (1 - axis);
                for (int d0 = d0_from; d0 <= d0_to; ++d0) {
                    const float slope = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
                    const float d1_cross = slope * ((float)d0 - p[0][0]) + p[0][1];
                    /* far off-screen (or NaN) crossings can never index a pixel */
                    if (!(d1_cross > -8.0f && d1_cross < (float)is + 8.0f)) continue;
                    int d1_in = (dir > 0) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
                    int d1_out = d1_in + dir;
                    if (d1_in < 0 || is <= d1_in) continue;
                    if (d1_out < 0 || is <= d1_out) continue;
                    const long base = (axis == 0) ? (long)d0 : (long)d0 * is;
                    const float alpha_in = idx[base + (long)d1_in * step] >= 0 ? 1.0f : 0.0f;
                    const float alpha_out = idx[base + (long)d1_out * step] >= 0 ? 1.0f : 0.0f;
                    const int in_is_fn = idx[base + (long)d1_in * step] == fn;
                    const float num = p[1][0] - p[0][0];
                    const int use0 = p[1][0] != (float)d0;
                    const int use1 = p[0][0] != (float)d0;
                    /* ---- outward sweep: from the pixel just outside the edge to the border */
                    if (in_is_fn) {
                        const int lim = (dir > 0) ? is - 1 : 0;
                        const int from = imax(imin(d1_out, lim), 0);
                        const int to = imin(imax(d1_out, lim), is - 1);
                        for (int d1 = from; d1 <= to; ++d1) {
                            const long q = base + (long)d1 * step;
                            const float a = idx[q] >= 0 ? 1.0f : 0.0f;
                            const float diff = (a - alpha_in) * ga[q];
                            if (!(diff > 0.0f)) continue;
                            if (use0) {
                                float dist = num / (p[1][0] - (float)d0) * ((float)d1 - d1_cross) * 2.0f / (float)is;
                                dist = (0.0f < dist) ? dist + eps : dist - eps;
                                g[gi0] -= diff / dist;
                            }
                            if (use1) {
                                float dist = num / ((float)d0 - p[0][0]) * ((float)d1 - d1_cross) * 2.0f / (float)is;
                                dist = (0.0f < dist) ? dist + eps : dist - eps;
                                g[gi1] -= diff / dist;
                            }
                        }
                    }
                    /* ---- inward sweep: from the pixel just inside to the opposite edge */
                    {
                        float c2;
                        if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0.0f)
                            c2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                        else
                            c2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                        if (!(c2 == c2)) continue; /* NaN from a degenerate opposite edge */
                        c2 = fminf(fmaxf(c2, -4.0f), (float)is + 4.0f);
                        const int lim = (dir > 0) ? (int)ceilf(c2) : (int)floorf(c2);
                        const int from = imax(imin(d1_in, lim), 0);
                        const int to = imin(imax(d1_in, lim), is - 1);
                        for (int d1 = from; d1 <= to; ++d1) {
                            const long q = base + (long)d1 * step;
                            if (idx[q] != fn) continue;
                            const float diff = (1.0f - alpha_out) * ga[q];
                            if (!(diff > 0.0f)) continue;
                            if (use0) {
                                float dist = num / (p[1][0] - (float)d0) * ((float)d1 - d1_cross) * 2.0f / (float)is;
                                dist = (0.0f < dist) ? dist + eps : dist - eps;
                                g[gi0] -= diff / dist;
                            }
                            if (use1) {
                                float dist = num / ((float)d0 - p[0][0]) * ((float)d1 - d1_cross) * 2.0f / (float)is;
                                dist = (0.0f < dist) ? dist + eps : dist - eps;
                                g[gi1] -= diff / dist;
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; ++k) out[k] = g[k];
    }
}

/*
 * Backward of the depth map w.r.t. the NDC face vertices (the NMR rasteriser's
 * backward_depth_map kernel, reached from reference homan/homan.py:391,406 when
 * the depth image carries a gradient).  Per covered sample: with zp its depth
 * and w_k its clamped, renormalised barycentrics,
 *     d z_k      +=  g * w_k * zp^2 / z_k^2
 *     d (x,y)_k  += -g * w_k * zp^2 * tmp[l] * is / 2,  tmp[l] = -sum_m inv[m][l] / z_m.
 * grad_depth is dL/d depth_map on the same `is` grid; grad_faces (B,NF,9) is
 * ACCUMULATED into (so it can follow orc_nmr_grad_faces_alpha).
 */
void orc_nmr_grad_faces_depth(const float *faces, const int32_t *idx_map,
                              const float *grad_depth, int B, int NF, int is,
                              float *grad_faces)
{
    const long npix = (long)is * is;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        for (int yi = 0; yi < is; ++yi)
            for (int xi = 0; xi < is; ++xi) {
                const long pix = b * npix + (long)yi * is + xi;
                const int fn = idx_map[pix];
                if (fn < 0) continue;
                const float g = grad_depth[pix];
                if (g == 0.0f) continue;
                const float *f = faces + ((long)b * NF + fn) * 9;
                float *gf = grad_faces + ((long)b * NF + fn) * 9;
                float p[3][2];
                for (int k = 0; k < 3; ++k)
                    for (int d = 0; d < 2; ++d) p[k][d] = orc_topix(f[3 * k + d], is);
                float inv[9] = {
                    p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                    p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                    p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
                const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                                  p[1][0] * (p[2][1] - p[0][1]);
                for (int k = 0; k < 9; ++k) inv[k] /= den;
                float w[3], ws = 0.0f;
                for (int k = 0; k < 3; ++k) {
                    float t = inv[3 * k + 0] * (float)xi;
                    t = t + inv[3 * k + 1] * (float)yi;
                    t = t + inv[3 * k + 2];
                    t = fminf(fmaxf(t, 0.0f), 1.0f);
                    w[k] = t;
                    ws += t;
                }
                float s = w[0] / f[2];
                s = s + w[1] / f[5];
                s = s + w[2] / f[8];
                const float zp = ws / s;
                const float zp2 = zp * zp;
                float tmp[2] = {0.0f, 0.0f};
                for (int l = 0; l < 2; ++l)
                    for (int m = 0; m < 3; ++m) tmp[l] += -inv[3 * m + l] / f[3 * m + 2];
                for (int k = 0; k < 3; ++k) {
                    const float wk = w[k] / ws;
                    const float zk = f[3 * k + 2];
                    gf[3 * k + 2] += g * wk * zp2 / (zk * zk);
                    for (int l = 0; l < 2; ++l)
                        gf[3 * k + l] += -g * tmp[l] * wk * zp2 * (float)is / 2.0f;
                }
            }
    }
}
