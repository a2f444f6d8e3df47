/*
 * oracle/csrc/sdf.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32, build with -ffp-contract=off) of the voxel
 * signed-distance field the reference gets from the third-party CUDA package
 * `sdf` (hassony2/multiperson @ HEAD, un-vendored, reference README.md:57-60):
 *   reference call sites: homan/interactions/scenesdf.py:9,32,119
 *       phi = SDF()(faces int32 (F,3), verts (B,V,3) in [-1,1]^3)  -> (B,32,32,32)
 *   consumed by scenesdf.py:120-123 (clamp >= 0) and :139-141 (grid_sample,
 *   x -> last grid dim).
 * The package source is NOT in /root/reference: PARITY UNPINNED.  The
 * algorithm restated here is the published voxel SDF of SDFGen (Bridson /
 * Batty): unsigned point-triangle distance (barycentric closest point with
 * edge clamping), sign from ray-crossing parity.
 *
 * Conventions fixed here (and mirrored by the HIP product kernels):
 *   - phi is indexed [b][k][j][i] with i <-> x fastest, j <-> y, k <-> z, so
 *     torch grid_sample(phi[:,None], grid(x,y,z)) addresses it directly.
 *   - voxel centres c(i) = -1 + (i + 0.5) * 2/N   (grid_sample
 *     align_corners=False geometry, the torch>=1.3 default the reference
 *     runs with, environment.yml:15).
 *   - phi > 0 inside, < 0 outside (scenesdf.py:120-122 "keep only inside").
 *   - inside iff an odd number of triangles is crossed by the ray from the
 *     voxel centre towards +x; a crossing is counted when the (y,z) projection
 *     of the centre lies in the projected triangle under the half-open
 *     orientation rule below and the hit abscissa is > the centre's x.
 */
#include <math.h>
#include <stdint.h>

static inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static float seg_dist(const float *x0, const float *x1, const float *x2)
{
    float dx[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
    float m2 = dot3(dx, dx);
    float e[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    float s12 = (m2 > 0.0f) ? dot3(e, dx) / m2 : 0.0f;
    if (s12 < 0.0f) s12 = 0.0f; else if (s12 > 1.0f) s12 = 1.0f;
    float t = 1.0f - s12;
    float c[3] = {s12 * x1[0] + t * x2[0], s12 * x1[1] + t * x2[1], s12 * x1[2] + t * x2[2]};
    float d[3] = {x0[0] - c[0], x0[1] - c[1], x0[2] - c[2]};
    return sqrtf(dot3(d, d));
}

float orc_point_triangle_distance(const float *x0, const float *x1, const float *x2, const float *x3)
{
    float x13[3] = {x1[0] - x3[0], x1[1] - x3[1], x1[2] - x3[2]};
    float x23[3] = {x2[0] - x3[0], x2[1] - x3[1], x2[2] - x3[2]};
    float x03[3] = {x0[0] - x3[0], x0[1] - x3[1], x0[2] - x3[2]};
    float m13 = dot3(x13, x13), m23 = dot3(x23, x23), d = dot3(x13, x23);
    float invdet = 1.0f / fmaxf(m13 * m23 - d * d, 1e-30f);
    float a = dot3(x13, x03), b = dot3(x23, x03);
    float w23 = invdet * (m23 * a - d * b);
    float w31 = invdet * (m13 * b - d * a);
    float w12 = 1.0f - w23 - w31;
    if (w23 >= 0.0f && w31 >= 0.0f && w12 >= 0.0f) {
        float c[3] = {w23 * x1[0] + w31 * x2[0] + w12 * x3[0],
                      w23 * x1[1] + w31 * x2[1] + w12 * x3[1],
                      w23 * x1[2] + w31 * x2[2] + w12 * x3[2]};
        float e[3] = {x0[0] - c[0], x0[1] - c[1], x0[2] - c[2]};
        return sqrtf(dot3(e, e));
    }
    if (w23 > 0.0f) return fminf(seg_dist(x0, x1, x2), seg_dist(x0, x1, x3));
    if (w31 > 0.0f) return fminf(seg_dist(x0, x1, x2), seg_dist(x0, x2, x3));
    return fminf(seg_dist(x0, x1, x3), seg_dist(x0, x2, x3));
}

/* sign of twice the signed area of (0,0),(x1,y1),(x2,y2) with a total tie-break */
static int orient2(float x1, float y1, float x2, float y2, float *tsa)
{
    *tsa = y1 * x2 - x1 * y2;
    if (*tsa > 0.0f) return 1;
    if (*tsa < 0.0f) return -1;
    if (y2 > y1) return 1;
    if (y2 < y1) return -1;
    if (x1 > x2) return 1;
    if (x1 < x2) return -1;
    return 0;
}

/* does the +x ray from c cross triangle (v1,v2,v3)?  *xhit = abscissa of the hit */
int orc_ray_x_crosses(const float *c, const float *v1, const float *v2, const float *v3, float *xhit)
{
    /* work in the (y,z) plane relative to c */
    float y1 = v1[1] - c[1], z1 = v1[2] - c[2];
    float y2 = v2[1] - c[1], z2 = v2[2] - c[2];
    float y3 = v3[1] - c[1], z3 = v3[2] - c[2];
    float a, b, g;
    int sa = orient2(y2, z2, y3, z3, &a);
    if (sa == 0) return 0;
    int sb = orient2(y3, z3, y1, z1, &b);
    if (sb != sa) return 0;
    int sc = orient2(y1, z1, y2, z2, &g);
    if (sc != sa) return 0;
    float sum = a + b + g;
    if (sum == 0.0f) return 0;
    float fa = a / sum, fb = b / sum, fc = g / sum;
    *xhit = fa * v1[0] + fb * v2[0] + fc * v3[0];
    return 1;
}

/*
 * phi (B,N,N,N).  clamp_outside != 0: voxels found outside get 0 and their
 * distance is not evaluated (equivalent after the reference's clamp(0),
 * scenesdf.py:121; used by the timed CPU baseline only).
 */
void orc_sdf_grid(const int32_t *faces, const float *verts, int B, int V, int F, int N,
                  int clamp_outside, float *phi)
{
    const float h = 2.0f / (float)N;
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
    for (int b = 0; b < B; ++b) {
        for (int kj = 0; kj < N * N; ++kj) {
            const int k = kj / N, j = kj % N;
            const float *vb = verts + (long)b * V * 3;
            float cy = -1.0f + ((float)j + 0.5f) * h;
            float cz = -1.0f + ((float)k + 0.5f) * h;
            /* crossing parity for the whole x-row at once */
            int cnt[64];
            for (int i = 0; i < N; ++i) cnt[i] = 0;
            for (int f = 0; f < F; ++f) {
                const float *v1 = vb + 3 * faces[3 * f + 0];
                const float *v2 = vb + 3 * faces[3 * f + 1];
                const float *v3 = vb + 3 * faces[3 * f + 2];
                float c[3] = {0.0f, cy, cz}, xh;
                /* a crossing needs the centre inside the (y,z) projection, hence inside its box */
                if (cy < fminf(v1[1], fminf(v2[1], v3[1])) || cy > fmaxf(v1[1], fmaxf(v2[1], v3[1])) ||
                    cz < fminf(v1[2], fminf(v2[2], v3[2])) || cz > fmaxf(v1[2], fmaxf(v2[2], v3[2])))
                    continue;
                if (!orc_ray_x_crosses(c, v1, v2, v3, &xh)) continue;
                for (int i = 0; i < N; ++i) {
                    float cx = -1.0f + ((float)i + 0.5f) * h;
                    if (xh > cx) cnt[i]++;
                }
            }
            for (int i = 0; i < N; ++i) {
                const int inside = cnt[i] & 1;
                float *o = phi + (((long)b * N + k) * N + j) * N + i;
                if (!inside && clamp_outside) { *o = 0.0f; continue; }
                float c[3] = {-1.0f + ((float)i + 0.5f) * h, cy, cz};
                float dmin = 1e30f;
                for (int f = 0; f < F; ++f) {
                    float d = orc_point_triangle_distance(c, vb + 3 * faces[3 * f + 0],
                                                          vb + 3 * faces[3 * f + 1],
                                                          vb + 3 * faces[3 * f + 2]);
                    if (d < dmin) dmin = d;
                }
                *o = inside ? dmin : -dmin;
            }
        }
    }
}
