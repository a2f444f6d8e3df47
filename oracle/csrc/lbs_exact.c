/*
 * oracle/csrc/lbs_exact.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The MANO layer's forward (pose from PCA coefficients, Rodrigues, shape / pose blend shapes, joint regression, kinematic
 * chain, rest-pose removal, linear blend skinning: the published smplx / MANO algorithm that oracle/lbs.py restates with
 * torch) written out operation by operation in ONE evaluation order, so that the hand's vertices are a defined function of
 * the parameters - the order csrc/mano.hip uses, hence bit-equal with the HIP kernels.  Same mathematics as oracle/lbs.py
 * (tests/test_objchain.py: within fp32 rounding of it); what fp32 leaves open is fixed as follows:
 *   - every sum runs sequentially from 0 in ascending index (PCA 16 terms, shape 10, skinning 16 joints); the 145 blend rows
 *     as four partial sums of 37 rows, combined as (p0 + p1) + (p2 + p3), then added to the template;
 *   - the joint regressor is folded into J_template (16,3) + J_shapedirs (16,3,10) (formed in double, rounded once);
 *   - sin / cos by oc_sincos below (argument reduction + fdlibm kernel polynomials in double, rounded to fp32);
 *   - 3x3 / 3-vector products left to right, no fused multiply-add (build with -ffp-contract=off).
 * Reference call sites of the layer: homan/manomodel.py:84-151; homan/homan.py:341-358.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define NV 778
#define NJ 16
#define NF 145
#define ROWS_PER_PART 37

static void oc_sincos(float af, float *sn, float *cs)
{
    const double a = (double)af;
    const double k = rint(a * 0.63661977236758138243);
    double r = a - k * 1.57079632673412561417e+00;
    r = r - k * 6.07710050650619224932e-11;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = -2.50507602534068634195e-08 + z * ps;
    ps = 2.75573137070700676789e-06 + z * ps;
    ps = -1.98412698298579493134e-04 + z * ps;
    ps = 8.33333333332248946124e-03 + z * ps;
    ps = -1.66666666666666324348e-01 + z * ps;
    const double s = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = 2.08757232129817482790e-09 + z * pc;
    pc = -2.75573143513906633035e-07 + z * pc;
    pc = 2.48015872894767294178e-05 + z * pc;
    pc = -1.38888888888741095749e-03 + z * pc;
    pc = 4.16666666666666019037e-02 + z * pc;
    const double c = (1.0 - 0.5 * z) + (z * z) * pc;
    const int q = (int)((long long)k) & 3;
    const double sv = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    const double cv = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    *sn = (float)sv;
    *cs = (float)cv;
}
void orc_sincos(float a, float *sn, float *cs) { oc_sincos(a, sn, cs); }

static void oc_rodrigues(const float *r, float *R)
{
    const float e0 = r[0] + 1e-8f, e1 = r[1] + 1e-8f, e2 = r[2] + 1e-8f;
    const float a = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
    const float nx = r[0] / a, ny = r[1] / a, nz = r[2] / a;
    float s, cs_;
    oc_sincos(a, &s, &cs_);
    const float c1 = 1.0f - cs_;
    const float K[9] = {0.f, -nz, ny, nz, 0.f, -nx, -ny, nx, 0.f};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float kk = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
            R[3 * i + j] = (i == j ? 1.0f : 0.0f) + s * K[3 * i + j] + c1 * kk;
        }
}

/* chain state of one frame: what csrc/mano.hip keeps in ManoShared */
typedef struct {
    float pose[48], Rl[NJ][9], J[NJ][3], Rw[NJ][9], tw[NJ][3], A[NJ][12], feat[NF];
    int depth[NJ], maxd;
} OcManoState;

typedef struct {
    const float *v_template, *M, *J_template, *J_shapedirs, *weights, *comps, *hand_mean;
    const int32_t *parents;
} OcManoModel;

static void oc_mano_prepare(const OcManoModel *m, const float *pca, int pca_stride, const float *rot, const float *betas, int b,
                            OcManoState *st)
{
    const int32_t *parents = m->parents;
    for (int t = 0; t < 48; ++t) {
        float v;
        if (t < 3) v = rot[b * 3 + t];
        else {
            const int k = t - 3;
            v = 0.f;
            for (int i = 0; i < 16; ++i) v += pca[(long)b * pca_stride + i] * m->comps[i * 45 + k];
            v += m->hand_mean[k];
        }
        st->pose[t] = v;
    }
    for (int t = 0; t < 10; ++t) st->feat[135 + t] = betas[b * 10 + t];
    int maxd = 0;
    for (int t = 0; t < NJ; ++t) {
        int d = 0;
        for (int q = parents[t]; q >= 0; q = parents[q]) ++d;
        st->depth[t] = d;
        if (d > maxd) maxd = d;
        float R[9];
        oc_rodrigues(&st->pose[3 * t], R);
        for (int k = 0; k < 9; ++k) st->Rl[t][k] = R[k];
        if (t >= 1)
            for (int k = 0; k < 9; ++k) st->feat[9 * (t - 1) + k] = R[k] - ((k % 4 == 0) ? 1.0f : 0.0f);
        for (int c = 0; c < 3; ++c) {
            float v = m->J_template[t * 3 + c];
            for (int l = 0; l < 10; ++l) v += betas[b * 10 + l] * m->J_shapedirs[(t * 3 + c) * 10 + l];
            st->J[t][c] = v;
        }
    }
    st->maxd = maxd;
    for (int level = 0; level <= maxd; ++level)
        for (int j = 0; j < NJ; ++j) {
            if (st->depth[j] != level) continue;
            const int p = parents[j];
            if (p < 0) {
                for (int k = 0; k < 9; ++k) st->Rw[j][k] = st->Rl[j][k];
                for (int c = 0; c < 3; ++c) st->tw[j][c] = st->J[j][c];
            } else {
                const float rel[3] = {st->J[j][0] - st->J[p][0], st->J[j][1] - st->J[p][1], st->J[j][2] - st->J[p][2]};
                for (int i = 0; i < 3; ++i) {
                    for (int k = 0; k < 3; ++k)
                        st->Rw[j][3 * i + k] = st->Rw[p][3 * i] * st->Rl[j][k] + st->Rw[p][3 * i + 1] * st->Rl[j][3 + k] +
                                               st->Rw[p][3 * i + 2] * st->Rl[j][6 + k];
                    st->tw[j][i] = st->Rw[p][3 * i] * rel[0] + st->Rw[p][3 * i + 1] * rel[1] + st->Rw[p][3 * i + 2] * rel[2] +
                                   st->tw[p][i];
                }
            }
        }
    for (int j = 0; j < NJ; ++j)
        for (int i = 0; i < 3; ++i) {
            st->A[j][4 * i] = st->Rw[j][3 * i];
            st->A[j][4 * i + 1] = st->Rw[j][3 * i + 1];
            st->A[j][4 * i + 2] = st->Rw[j][3 * i + 2];
            st->A[j][4 * i + 3] = st->tw[j][i] - (st->Rw[j][3 * i] * st->J[j][0] + st->Rw[j][3 * i + 1] * st->J[j][1] +
                                                  st->Rw[j][3 * i + 2] * st->J[j][2]);
        }
}

/* blended rest-pose vertex v (template + shape / pose blend shapes) */
static void oc_mano_vp(const OcManoModel *m, const OcManoState *st, int v, float *vp)
{
    for (int c = 0; c < 3; ++c) {
        float part[4];
        for (int w = 0; w < 4; ++w) {
            const int k0 = w * ROWS_PER_PART, k1 = (k0 + ROWS_PER_PART < NF) ? k0 + ROWS_PER_PART : NF;
            float a = 0.f;
            for (int k = k0; k < k1; ++k) a += st->feat[k] * m->M[(long)k * (3 * NV) + 3 * v + c];
            part[w] = a;
        }
        vp[c] = m->v_template[3 * v + c] + ((part[0] + part[1]) + (part[2] + part[3]));
    }
}

static void oc_mano_skin(const OcManoModel *m, const OcManoState *st, int v, float *T)
{
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
    for (int j = 0; j < NJ; ++j)
        for (int k = 0; k < 12; ++k) T[k] += m->weights[v * NJ + j] * st->A[j][k];
}

/*
 * pca (B, pca_stride) [first 16 columns used], rot (B,3), betas (B,10) -> verts (B,778,3) WITHOUT the model-space translation
 * (the caller adds mano_trans: the kernel's last addition).  Model arrays in the kernel's layout (homan_amd.mano_assets.
 * kernel_layout): v_template (778,3), M (145,2334) = [posedirs ; shapedirs^T], J_template (16,3), J_shapedirs (16,3,10),
 * weights (778,16), comps (16,45), hand_mean (45), parents (16).
 */
void orc_mano_forward(const float *v_template, const float *M, const float *J_template, const float *J_shapedirs,
                      const float *weights, const float *comps, const float *hand_mean, const int32_t *parents,
                      const float *pca, int pca_stride, const float *rot, const float *betas, int B, float *verts)
{
    const OcManoModel m = {v_template, M, J_template, J_shapedirs, weights, comps, hand_mean, parents};
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        OcManoState st;
        oc_mano_prepare(&m, pca, pca_stride, rot, betas, b, &st);
        for (int v = 0; v < NV; ++v) {
            float vp[3], T[12];
            oc_mano_vp(&m, &st, v, vp);
            oc_mano_skin(&m, &st, v, T);
            float *o = verts + ((long)b * NV + v) * 3;
            for (int i = 0; i < 3; ++i) o[i] = T[4 * i] * vp[0] + T[4 * i + 1] * vp[1] + T[4 * i + 2] * vp[2] + T[4 * i + 3];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * The HAND's gradient chain of one optimisation step, written out in the evaluation order of csrc/mano.hip's backward
 * (k_mano_bwd<RIGID> + mano_bwd2_body): the same mathematics as autograd through oracle/lbs.py and transform_persp (reference
 * homan/homan.py:341-382, manomodel.py:84-151), with every reduction in ONE stated order:
 *   - vertices in 13 chunks of 64; inside a chunk a sum over the vertices is either sequential (the 16 x 12 skinning sums) or the
 *     six-step pairwise tree of oc_wave_sum (DPP tree of hm_wave_sum: quads, rows of 16, rows 0-1 / 2-3, halves);
 *   - the chunk results are added sequentially, chunk 0 first;
 *   - kinematic chain leaves first, a parent collecting its children in ascending joint index.
 * ---------------------------------------------------------------------------------------------------------------------- */
void oc_rot6d_to_mat(const float *r6, float *R);                 /* objchain.c */
void oc_rot6d_backward(const float *r6, const float *dR, float *dr6);

static float oc_wave_sum(const float *a)      /* 64 values -> the value hm_wave_sum leaves in lane 63 */
{
    float R[4];
    for (int r = 0; r < 4; ++r) {
        float Q[4];
        for (int q = 0; q < 4; ++q) {
            const float *x = a + 16 * r + 4 * q;
            Q[q] = (x[3] + x[2]) + (x[1] + x[0]);
        }
        R[r] = (Q[3] + Q[2]) + (Q[1] + Q[0]);
    }
    return (R[3] + R[2]) + (R[1] + R[0]);
}
/* hm_block_sum over a 256-thread workgroup whose waves 1..3 hold zeros */
static float oc_block_sum64(const float *a)
{
    const float z[64] = {0.f};
    float t = 0.f;
    t += oc_wave_sum(a);
    for (int w = 1; w < 4; ++w) t += oc_wave_sum(z);
    return t;
}

static void oc_rodrigues_backward(const float *r, const float *dR, float *dr)
{
    const float e[3] = {r[0] + 1e-8f, r[1] + 1e-8f, r[2] + 1e-8f};
    const float a = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    const float n[3] = {r[0] / a, r[1] / a, r[2] / a};
    float s, c;
    oc_sincos(a, &s, &c);
    const float c1 = 1.0f - c;
    const float K[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
    float KK[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KK[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    float dRK = 0.f, dRKK = 0.f;
    for (int k = 0; k < 9; ++k) { dRK += dR[k] * K[k]; dRKK += dR[k] * KK[k]; }
    float da = c * dRK + s * dRKK;
    float dK[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float t1 = dR[3 * i] * K[3 * j] + dR[3 * i + 1] * K[3 * j + 1] + dR[3 * i + 2] * K[3 * j + 2];
            float t2 = K[i] * dR[j] + K[3 + i] * dR[3 + j] + K[6 + i] * dR[6 + j];
            dK[3 * i + j] = s * dR[3 * i + j] + c1 * (t1 + t2);
        }
    const float dn[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    da += -(dn[0] * r[0] + dn[1] * r[1] + dn[2] * r[2]) / (a * a);
    for (int i = 0; i < 3; ++i) dr[i] = dn[i] / a + da * e[i] / a;
}

#define NCH 13
#define PART 352
/*
 * mesh (B,778,3): the forward's model-space vertices (LBS + mano_trans); rot6d (B,6), scale: the hand's rigid pose;
 * terms[k] (B,778,3) with weights tw[k]: gradient terms on the camera-space vertices (they reach the mesh and the rigid pose);
 * g_frame (B, frame_stride) or NULL: one vector per frame, times frame_scale, that reaches the rigid pose only;
 * g_pca_extra (B, pca_stride) or NULL, times w_extra: added to the PCA gradient (the prior's unit gradient).
 * -> g_pca (B,pca_stride), g_rot (B,3), g_betas (B,10), g_trans (B,3) [MANO], g_rot6d (B,6), g_rtrans (B,3) [rigid]
 */
static void oc_hand_chain_rows(const float *v_template, const float *M, const float *J_template, const float *J_shapedirs,
                    const float *weights, const float *comps, const float *hand_mean, const int32_t *parents,
                    const float *pca, int pca_stride, const float *rot, const float *betas, const float *mesh,
                    const float *rot6d, float scale, const float *const *terms, const float *tw, int n_terms,
                    const float *g_frame, int frame_stride, float frame_scale, const float *g_pca_extra, float w_extra, int B,
                    float *g_pca, float *g_rot, float *g_betas, float *g_trans, float *g_rot6d, float *g_rtrans,
                    const float *gmesh, int row0, int row_stride, const float *g_rigid)
{
    const OcManoModel m = {v_template, M, J_template, J_shapedirs, weights, comps, hand_mean, parents};
#pragma omp parallel for schedule(static)
    for (int lb = 0; lb < B; ++lb) {
        const int b = lb * row_stride + row0;
        OcManoState st;
        oc_mano_prepare(&m, pca, pca_stride, rot, betas, b, &st);
        float R[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!gmesh) oc_rot6d_to_mat(rot6d + (long)b * 6, R);
        const float s = scale;
        static const float zero64[64] = {0.f};
        float tot[PART];
        float part[NCH][PART];
        for (int ch = 0; ch < NCH; ++ch) {
            const int v0 = ch * 64, nv = (NV - v0 < 64) ? NV - v0 : 64;
            float racc[12][64], g[3][64], s_g[64][3], s_vp[64][3], s_dvp[192], s_w[64][NJ];
            for (int k = 0; k < 12; ++k) for (int t = 0; t < 64; ++t) racc[k][t] = 0.f;
            for (int c = 0; c < 3; ++c) for (int t = 0; t < 64; ++t) g[c][t] = 0.f;
            for (int t = 0; t < 192; ++t) s_dvp[t] = 0.f;
            for (int t = 0; t < nv; ++t) {
                const int v = v0 + t;
                const long o = ((long)b * NV + v) * 3;
                if (gmesh) {        /* the model-space gradient is handed in (csrc/mano.hip k_mano_bwd<false>) */
                    g[0][t] = gmesh[o]; g[1][t] = gmesh[o + 1]; g[2][t] = gmesh[o + 2];
                    float T[12];
                    oc_mano_skin(&m, &st, v, T);
                    oc_mano_vp(&m, &st, v, s_vp[t]);
                    for (int c = 0; c < 3; ++c) {
                        s_g[t][c] = g[c][t];
                        s_dvp[3 * t + c] = T[c] * g[0][t] + T[4 + c] * g[1][t] + T[8 + c] * g[2][t];
                    }
                    for (int j = 0; j < NJ; ++j) s_w[t][j] = weights[v * NJ + j];
                    continue;
                }
                const float mv[3] = {mesh[o], mesh[o + 1], mesh[o + 2]};
                float gf[3] = {0.f, 0.f, 0.f}, gt[3];
                for (int k = 0; k < n_terms; ++k) {
                    gf[0] += tw[k] * terms[k][o]; gf[1] += tw[k] * terms[k][o + 1]; gf[2] += tw[k] * terms[k][o + 2];
                }
                for (int c = 0; c < 3; ++c)
                    gt[c] = gf[c] + (g_frame ? frame_scale * g_frame[(long)b * frame_stride + c] : 0.f);
                if (g_rigid) { gt[0] += g_rigid[o]; gt[1] += g_rigid[o + 1]; gt[2] += g_rigid[o + 2]; }   /* reaches R, t only */
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) racc[3 * i + j][t] = (s * mv[i]) * gt[j];
                for (int j = 0; j < 3; ++j) racc[9 + j][t] = gt[j];
                g[0][t] = s * (R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2]);
                g[1][t] = s * (R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2]);
                g[2][t] = s * (R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]);
                float T[12];
                oc_mano_skin(&m, &st, v, T);
                oc_mano_vp(&m, &st, v, s_vp[t]);
                for (int c = 0; c < 3; ++c) {
                    s_g[t][c] = g[c][t];
                    s_dvp[3 * t + c] = T[c] * g[0][t] + T[4 + c] * g[1][t] + T[8 + c] * g[2][t];
                }
                for (int j = 0; j < NJ; ++j) s_w[t][j] = weights[v * NJ + j];
            }
            float *out = part[ch];
            for (int k = 0; k < 12; ++k) out[340 + k] = oc_block_sum64(racc[k]);
            for (int c = 0; c < 3; ++c) out[337 + c] = oc_block_sum64(g[c]);
            for (int t = 0; t < 192; ++t) {
                const int j = t / 12, r = (t % 12) / 4, c = t % 4;
                float acc = 0.f;
                for (int i = 0; i < nv; ++i) acc += s_w[i][j] * s_g[i][r] * (c < 3 ? s_vp[i][c] : 1.0f);
                out[t] = acc;
            }
            const int ne = 3 * nv;
            for (int k = 0; k < NF; ++k) {
                const float *row = M + (long)k * (3 * NV) + 3 * v0;
                float acc[64];
                for (int lane = 0; lane < 64; ++lane) {
                    const float r0 = lane < ne ? row[lane] : 0.f, r1 = lane + 64 < ne ? row[lane + 64] : 0.f;
                    const float r2 = lane + 128 < ne ? row[lane + 128] : 0.f;
                    const float d0 = lane < ne ? s_dvp[lane] : 0.f, d1 = lane + 64 < ne ? s_dvp[lane + 64] : 0.f;
                    const float d2 = lane + 128 < ne ? s_dvp[lane + 128] : 0.f;
                    float a = r0 * d0;
                    a += r1 * d1;
                    a += r2 * d2;
                    acc[lane] = a;
                }
                out[192 + k] = oc_wave_sum(acc);
            }
            (void)zero64;
        }
        for (int k = 0; k < PART; ++k) {
            float a = 0.f;
            for (int c = 0; c < NCH; ++c) a += part[c][k];
            tot[k] = a;
        }
        /* second half: chain / Rodrigues / PCA backward */
        float dRw[NJ][9], dtw[NJ][3], dJ[NJ][3], dRl[NJ][9], dpose[48], cR[NJ][9], ct[NJ][3], cJ[NJ][3];
        for (int j = 0; j < NJ; ++j) {
            const float *dA = &tot[j * 12];
            for (int i = 0; i < 3; ++i) {
                for (int k = 0; k < 3; ++k) dRw[j][3 * i + k] = dA[4 * i + k] - dA[4 * i + 3] * st.J[j][k];
                dtw[j][i] = dA[4 * i + 3];
            }
            for (int k = 0; k < 3; ++k)
                dJ[j][k] = -(st.Rw[j][k] * dA[3] + st.Rw[j][3 + k] * dA[7] + st.Rw[j][6 + k] * dA[11]);
        }
        for (int level = st.maxd; level >= 1; --level) {
            for (int j = 0; j < NJ; ++j) {
                if (st.depth[j] != level) continue;
                const int p = parents[j];
                const float rel[3] = {st.J[j][0] - st.J[p][0], st.J[j][1] - st.J[p][1], st.J[j][2] - st.J[p][2]};
                for (int i = 0; i < 3; ++i) {
                    for (int k = 0; k < 3; ++k)
                        cR[j][3 * i + k] = dtw[j][i] * rel[k] + dRw[j][3 * i] * st.Rl[j][3 * k] +
                                           dRw[j][3 * i + 1] * st.Rl[j][3 * k + 1] + dRw[j][3 * i + 2] * st.Rl[j][3 * k + 2];
                    ct[j][i] = dtw[j][i];
                }
                for (int k = 0; k < 3; ++k) {
                    const float d = st.Rw[p][k] * dtw[j][0] + st.Rw[p][3 + k] * dtw[j][1] + st.Rw[p][6 + k] * dtw[j][2];
                    dJ[j][k] += d;
                    cJ[j][k] = -d;
                }
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < 3; ++k)
                        dRl[j][3 * i + k] = st.Rw[p][i] * dRw[j][k] + st.Rw[p][3 + i] * dRw[j][3 + k] + st.Rw[p][6 + i] * dRw[j][6 + k];
            }
            for (int t = 0; t < NJ; ++t) {
                if (st.depth[t] != level - 1) continue;
                for (int j = 0; j < NJ; ++j)
                    if (parents[j] == t) {
                        for (int k = 0; k < 9; ++k) dRw[t][k] += cR[j][k];
                        for (int k = 0; k < 3; ++k) { dtw[t][k] += ct[j][k]; dJ[t][k] += cJ[j][k]; }
                    }
            }
        }
        for (int t = 0; t < NJ; ++t)
            if (parents[t] < 0) {
                for (int k = 0; k < 9; ++k) dRl[t][k] = dRw[t][k];
                for (int c = 0; c < 3; ++c) dJ[t][c] += dtw[t][c];
            }
        for (int t = 0; t < NJ; ++t) {
            float dR[9], dr[3];
            for (int k = 0; k < 9; ++k) dR[k] = dRl[t][k] + (t >= 1 ? tot[192 + 9 * (t - 1) + k] : 0.f);
            oc_rodrigues_backward(&st.pose[3 * t], dR, dr);
            for (int c = 0; c < 3; ++c) dpose[3 * t + c] = dr[c];
        }
        for (int t = 0; t < 3; ++t) {
            g_rot[b * 3 + t] = dpose[t];
            g_trans[b * 3 + t] = tot[337 + t];
        }
        for (int i = 0; i < pca_stride; ++i) {
            float a = 0.f;
            if (i < 16)
                for (int k = 0; k < 45; ++k) a += comps[i * 45 + k] * dpose[3 + k];
            if (g_pca_extra) a += w_extra * g_pca_extra[(long)b * pca_stride + i];
            g_pca[(long)b * pca_stride + i] = a;
        }
        for (int t = 0; t < 10; ++t) {
            float a = tot[192 + 135 + t];
            for (int q = 0; q < NJ * 3; ++q) a += J_shapedirs[q * 10 + t] * dJ[q / 3][q % 3];
            g_betas[b * 10 + t] = a;
        }
        if (!gmesh) {
            float dr6[6];
            oc_rot6d_backward(rot6d + (long)b * 6, &tot[340], dr6);
            for (int k = 0; k < 6; ++k) g_rot6d[(long)b * 6 + k] = dr6[k];
            for (int k = 0; k < 3; ++k) g_rtrans[(long)b * 3 + k] = tot[349 + k];
        }
    }
}
void orc_hand_chain(const float *v_template, const float *M, const float *J_template, const float *J_shapedirs,
                    const float *weights, const float *comps, const float *hand_mean, const int32_t *parents,
                    const float *pca, int pca_stride, const float *rot, const float *betas, const float *mesh,
                    const float *rot6d, float scale, const float *const *terms, const float *tw, int n_terms,
                    const float *g_frame, int frame_stride, float frame_scale, const float *g_pca_extra, float w_extra, int B,
                    float *g_pca, float *g_rot, float *g_betas, float *g_trans, float *g_rot6d, float *g_rtrans)
{
    oc_hand_chain_rows(v_template, M, J_template, J_shapedirs, weights, comps, hand_mean, parents, pca, pca_stride, rot, betas, mesh,
                       rot6d, scale, terms, tw, n_terms, g_frame, frame_stride, frame_scale, g_pca_extra, w_extra, B, g_pca, g_rot,
                       g_betas, g_trans, g_rot6d, g_rtrans, NULL, 0, 1, NULL);
}
/* the same with one more per-vertex term g_rigid (B,778,3) that reaches the rigid pose only (inter_type "min", reference
 * homan/losses.py:219-221: the closest pair's pull on the mesh-detached hand) */
void orc_hand_chain_rigid(const float *v_template, const float *M, const float *J_template, const float *J_shapedirs,
                    const float *weights, const float *comps, const float *hand_mean, const int32_t *parents,
                    const float *pca, int pca_stride, const float *rot, const float *betas, const float *mesh,
                    const float *rot6d, float scale, const float *const *terms, const float *tw, int n_terms,
                    const float *g_rigid, const float *g_pca_extra, float w_extra, int B,
                    float *g_pca, float *g_rot, float *g_betas, float *g_trans, float *g_rot6d, float *g_rtrans)
{
    oc_hand_chain_rows(v_template, M, J_template, J_shapedirs, weights, comps, hand_mean, parents, pca, pca_stride, rot, betas, mesh,
                       rot6d, scale, terms, tw, n_terms, NULL, 0, 0.f, g_pca_extra, w_extra, B, g_pca, g_rot,
                       g_betas, g_trans, g_rot6d, g_rtrans, NULL, 0, 1, g_rigid);
}
/* The MANO layer's backward alone for the rows row0, row0 + row_stride, ... (B of them) of arrays holding B * row_stride rows
 * (two hands per frame are interleaved frame-major, reference homan/homan.py:62-63, each hand through its side's model): the
 * model-space vertex gradient gmesh (rows,778,3) is handed in (csrc/mano.hip k_mano_bwd<false>, hm_mano_bwd_rows). */
void orc_mano_bwd_rows(const float *v_template, const float *M, const float *J_template, const float *J_shapedirs,
                       const float *weights, const float *comps, const float *hand_mean, const int32_t *parents,
                       const float *pca, int pca_stride, const float *rot, const float *betas, const float *gmesh,
                       const float *g_pca_extra, float w_extra, int B, int row0, int row_stride, float *g_pca, float *g_rot,
                       float *g_betas, float *g_trans)
{
    oc_hand_chain_rows(v_template, M, J_template, J_shapedirs, weights, comps, hand_mean, parents, pca, pca_stride, rot, betas, NULL,
                       NULL, 1.0f, NULL, NULL, 0, NULL, 0, 0.f, g_pca_extra, w_extra, B, g_pca, g_rot, g_betas, g_trans, NULL, NULL,
                       gmesh, row0, row_stride, NULL);
}

/* The hands' rigid backward as a launch of its own (csrc/geometry.hip k_rigid_bwd<false>, one workgroup of `nthreads` threads per
 * row, V <= nthreads: thread v holds vertex v; 13 block sums - waves through oc_wave_sum, wave results in wave order).
 * mesh (N,V,3) model-space vertices, rot6d (N,6), one scale; terms / g_frame as in orc_hand_chain.
 * -> g_mesh (N,V,3), g_rot6d (N,6), g_trans (N,3) */
void orc_rigid_bwd_rows(const float *mesh, const float *rot6d, float scale, const float *const *terms, const float *tw, int n_terms,
                        const float *g_frame, int frame_stride, float frame_scale, int N, int V, int nthreads, float *g_mesh,
                        float *g_rot6d, float *g_trans)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float R[9];
        oc_rot6d_to_mat(rot6d + (long)n * 6, R);
        const float s = scale;
        float gfr[3] = {0.f, 0.f, 0.f};
        if (g_frame)
            for (int c = 0; c < 3; ++c) gfr[c] = frame_scale * g_frame[(long)n * frame_stride + c];
        float acc[13][1024];
        for (int k = 0; k < 13; ++k) for (int t = 0; t < nthreads; ++t) acc[k][t] = 0.f;
        for (int v = 0; v < V; ++v) {
            const long o = ((long)n * V + v) * 3;
            const float m[3] = {mesh[o], mesh[o + 1], mesh[o + 2]};
            float gf[3] = {0.f, 0.f, 0.f}, gt[3];
            for (int k = 0; k < n_terms; ++k) {
                gf[0] += tw[k] * terms[k][o]; gf[1] += tw[k] * terms[k][o + 1]; gf[2] += tw[k] * terms[k][o + 2];
            }
            gt[0] = gf[0] + gfr[0]; gt[1] = gf[1] + gfr[1]; gt[2] = gf[2] + gfr[2];
            const float dm[3] = {R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2], R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2],
                                 R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]};
            const int t = v % nthreads;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) acc[3 * i + j][t] += (s * m[i]) * gt[j];
            for (int j = 0; j < 3; ++j) acc[9 + j][t] += gt[j];
            acc[12][t] += m[0] * dm[0] + m[1] * dm[1] + m[2] * dm[2];
            g_mesh[o] = s * dm[0]; g_mesh[o + 1] = s * dm[1]; g_mesh[o + 2] = s * dm[2];
        }
        float tot[13];
        for (int k = 0; k < 13; ++k) {
            float a = 0.f;
            for (int w = 0; w < nthreads / 64; ++w) a += oc_wave_sum(acc[k] + 64 * w);
            tot[k] = a;
        }
        oc_rot6d_backward(rot6d + (long)n * 6, tot, g_rot6d + (long)n * 6);
        for (int k = 0; k < 3; ++k) g_trans[(long)n * 3 + k] = tot[9 + k];
    }
}

/* 2-D reprojection term's unit gradient on the camera-space hand vertices (reference homan/losses.py:141-164; csrc/pair_bodies.h
 * hand_terms_body): element-wise.  verts (N,V,3), K (N / hand_nb,3,3), ref2d (N,V,2) -> unit (N,V,3) */
void orc_v2d_unit_grad(const float *verts, const float *camintr, const float *ref2d, float image_size, int N, int V, int hand_nb,
                       float *unit)
{
    const long total = (long)N * V;
    const float inv_cnt = 1.0f / (float)total;
    for (long i = 0; i < total; ++i) {
        const float *k = camintr + ((i / V) / hand_nb) * 9;      /* rows are interleaved frame-major: one camera per frame */
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        const float hx = k[0] * x + k[1] * y + k[2] * z;
        const float hy = k[3] * x + k[4] * y + k[5] * z;
        const float hz = k[6] * x + k[7] * y + k[8] * z;
        const float px = hx / hz, py = hy / hz;
        const float rx = ref2d[2 * i], ry = ref2d[2 * i + 1];
        const float dx = px - rx / image_size, dy = py - ry / image_size;
        const float gpx = 2.0f * dx * inv_cnt, gpy = 2.0f * dy * inv_cnt;
        const float ghx = gpx / hz, ghy = gpy / hz, ghz = -(gpx * hx + gpy * hy) / (hz * hz);
        unit[3 * i] = k[0] * ghx + k[3] * ghy + k[6] * ghz;
        unit[3 * i + 1] = k[1] * ghx + k[4] * ghy + k[7] * ghz;
        unit[3 * i + 2] = k[2] * ghx + k[5] * ghy + k[8] * ghz;
    }
}

/* Coarse interaction term (reference homan/losses.py:199-242, utils/bbox.py:111-135, utils/geometry.py:69-86; csrc/pair_bodies.h
 * inter_body): per frame {gate, centroid MSE, gate * 2 (c_hand - c_obj) / 3}.  The centroid sums: `nthreads` strided partial sums
 * (thread t takes vertices t, t + nthreads, ...), 64 at a time through oc_wave_sum, the wave results added in wave order.
 * vh (B,Vh,3), vo (B,Vo,3), K (B,3,3) -> rec (B,8) [0]=gate [1]=mse [2..4]=vector */
void orc_inter_rec(const float *vh, const float *vo, const float *camintr, int B, int Vh, int Vo, float expansion, float zthresh,
                   int nthreads, float *rec)
{
    for (int b = 0; b < B; ++b) {
        const float *k = camintr + b * 9;
        float box[2][4], zr[2][2], cen[2][3];
        for (int which = 0; which < 2; ++which) {
            const float *v = which == 0 ? vo + (long)b * Vo * 3 : vh + (long)b * Vh * 3;
            const int V = which == 0 ? Vo : Vh;
            float umin = 3.4e38f, umax = -3.4e38f, vmin = 3.4e38f, vmax = -3.4e38f, zmin = 3.4e38f, zmax = -3.4e38f;
            float sx[1024], sy[1024], sz[1024];
            for (int t = 0; t < nthreads; ++t) {
                float ax = 0.f, ay = 0.f, az = 0.f;
                for (int i = t; i < V; i += nthreads) {
                    const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
                    const float zz = z + 1e-9f;
                    const float xn = x / zz, yn = (y * -1.0f) / zz;
                    float u = xn * k[0] + yn * k[1];
                    u = u + k[2];
                    float w = xn * k[3] + yn * k[4];
                    w = w + k[5];
                    w = 1.0f - w;
                    u = 2.0f * (u - 0.5f);
                    w = 2.0f * (w - 0.5f);
                    umin = fminf(umin, u); umax = fmaxf(umax, u);
                    vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
                    zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
                    ax += x; ay += y; az += z;
                }
                sx[t] = ax; sy[t] = ay; sz[t] = az;
            }
            float t9[9] = {umin, umax, vmin, vmax, zmin, zmax, 0.f, 0.f, 0.f};
            for (int w = 0; w < nthreads / 64; ++w) {
                t9[6] = t9[6] + oc_wave_sum(sx + 64 * w);
                t9[7] = t9[7] + oc_wave_sum(sy + 64 * w);
                t9[8] = t9[8] + oc_wave_sum(sz + 64 * w);
            }
            const float cx = (t9[0] + t9[1]) / 2.0f, cy = (t9[2] + t9[3]) / 2.0f;
            const float ex = (t9[1] - t9[0]) / 2.0f * (1.0f + expansion), ey = (t9[3] - t9[2]) / 2.0f * (1.0f + expansion);
            box[which][0] = cx - ex; box[which][1] = cy - ey; box[which][2] = cx + ex; box[which][3] = cy + ey;
            zr[which][0] = t9[4]; zr[which][1] = t9[5];
            cen[which][0] = t9[6] / (float)V; cen[which][1] = t9[7] / (float)V; cen[which][2] = t9[8] / (float)V;
        }
        const float a1 = (box[0][2] - box[0][0]) * (box[0][3] - box[0][1]);
        const float a2 = (box[1][2] - box[1][0]) * (box[1][3] - box[1][1]);
        const float w = fmaxf(fminf(box[0][2], box[1][2]) - fmaxf(box[0][0], box[1][0]), 0.f);
        const float h = fmaxf(fminf(box[0][3], box[1][3]) - fmaxf(box[0][1], box[1][1]), 0.f);
        const float inter = w * h;
        const float iou = inter / (a1 + a2 - inter);
        const float a = zr[0][0], bb = zr[0][1], c = zr[1][0], d = zr[1][1];
        const float zd = (d >= a && bb >= c) ? 0.f : fminf(fabsf(c - bb), fabsf(a - d));
        const float flag = ((iou > 0.f) && (zd < zthresh)) ? 1.f : 0.f;
        const float dx = cen[1][0] - cen[0][0], dy = cen[1][1] - cen[0][1], dz = cen[1][2] - cen[0][2];
        float *r = rec + b * 8;
        r[0] = flag;
        r[1] = (dx * dx + dy * dy + dz * dz) / 3.0f;
        r[2] = flag * 2.0f * dx / 3.0f; r[3] = flag * 2.0f * dy / 3.0f; r[4] = flag * 2.0f * dz / 3.0f;
        r[5] = r[6] = r[7] = 0.f;
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Step-2 terms (contact, collision) in the kernels' order: csrc/contact.hip, csrc/pair_bodies.h nn_full_body, csrc/sdf.hip.
 * ---------------------------------------------------------------------------------------------------------------------- */
static float oc_tanh(float xf)        /* csrc/hm_common.h hm_tanh, same operations */
{
    const double ax = xf < 0.f ? -(double)xf : (double)xf;
    if (ax > 20.0) return xf < 0.f ? -1.0f : 1.0f;
    if (ax < 0.01) {
        const double q = ax * ax;
        const double sm = ax * (1.0 + q * (-3.33333333333333314830e-01 + q * (1.33333333333333331483e-01 + q * -5.39682539682539708542e-02)));
        return (float)(xf < 0.f ? -sm : sm);
    }
    const double t = -2.0 * ax;
    const double k = rint(t * 1.44269504088896338700e+00);
    const double r = (t - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    const double z = r * r;
    double p = 4.13813679705723846039e-08;
    p = -1.65339022054652515390e-06 + z * p;
    p = 6.61375632143793436117e-05 + z * p;
    p = -2.77777777770155933842e-03 + z * p;
    p = 1.66666666666666019037e-01 + z * p;
    const double c = r - z * p;
    const double er = 1.0 - ((r * c) / (c - 2.0) - r);
    const double e = ldexp(er, (int)k);
    const double th = (1.0 - e) / (1.0 + e);
    return (float)(xf < 0.f ? -th : th);
}
float orc_tanh(float x) { return oc_tanh(x); }

/* nearest object vertex of every hand vertex (reference homan/interactions/contactloss.py:60-79, 162-163): squared distance
 * (ox-hx)^2 + (oy-hy)^2 + (oz-hz)^2 left to right, ties -> the lowest index.  -> idx (B,Vh) */
void orc_nn_search_d2(const float *vh, const float *vo, int B, int Vh, int Vo, int32_t *idx, float *d2);
void orc_nn_search(const float *vh, const float *vo, int B, int Vh, int Vo, int32_t *idx) { orc_nn_search_d2(vh, vo, B, Vh, Vo, idx, NULL); }
void orc_nn_search_d2(const float *vh, const float *vo, int B, int Vh, int Vo, int32_t *idx, float *d2)
{
#pragma omp parallel for schedule(static)
    for (long bi = 0; bi < (long)B * Vh; ++bi) {
        const int b = (int)(bi / Vh);
        const float *h = vh + bi * 3;
        float best = 3.4e38f;
        int besti = 0;
        for (int j = 0; j < Vo; ++j) {
            const float *o = vo + ((long)b * Vo + j) * 3;
            const float dx = o[0] - h[0], dy = o[1] - h[1], dz = o[2] - h[2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; besti = j; }
        }
        idx[bi] = besti;
        if (d2) d2[bi] = best;
    }
}

/* contact term as the reference executes it (contactloss.py:228-257 with an empty attraction mask: mean over the clip's
 * frames and hand vertices of thresh * tanh(|nn - h| / thresh)): unit gradient on the hand vertices, and on the object
 * vertices minus the sum of the gradients of the hand vertices that picked them - an exact sum in 2^-44 fixed point.
 * -> g_hand (B,Vh,3), g_obj (B,Vo,3) */
void orc_contact_grads(const float *vh, const float *vo, const int32_t *idx, int B, int Vh, int Vo, float thresh, int clip_len,
                       float *g_hand, float *g_obj)
{
    const float FIX = 17592186044416.0f;      /* 2^44 */
    const float inv_cnt = 1.0f / (float)((long)clip_len * Vh);
    for (int b = 0; b < B; ++b) {
        long long *acc = (long long *)calloc((size_t)Vo * 3, sizeof(long long));
        for (int i = 0; i < Vh; ++i) {
            const int j = idx[(long)b * Vh + i];
            const float *h = vh + ((long)b * Vh + i) * 3;
            const float *o = vo + ((long)b * Vo + j) * 3;
            const float dx = o[0] - h[0], dy = o[1] - h[1], dz = o[2] - h[2];
            const float a = sqrtf(dx * dx + dy * dy + dz * dz);
            const float th = oc_tanh(a / thresh);
            const float k = (a > 0.f) ? (1.0f - th * th) / a * inv_cnt : 0.f;
            const float g[3] = {-k * dx, -k * dy, -k * dz};
            for (int c = 0; c < 3; ++c) {
                g_hand[((long)b * Vh + i) * 3 + c] = g[c];
                acc[3 * j + c] += llrintf(g[c] * FIX);
            }
        }
        for (int i = 0; i < 3 * Vo; ++i) g_obj[(long)b * Vo * 3 + i] = -(float)acc[i] * (1.0f / FIX);
        free(acc);
    }
}

/* Collision term, one ordered pair (reference homan/interactions/scenesdf.py:124-146): d sum_i trilinear(phi, (p_i - c) / s) /
 * d p_i for the points p (B,V,3) against the clamped SDF grid phi (B,N,N,N; [z][y][x], zero outside the mesh and out of
 * bounds) of the mesh whose box is boxes (B,4) = centre xyz, scale.  align_corners = False.  Element-wise; csrc/sdf.hip
 * k_sdf_sample.  -> g (B,V,3) */
void orc_sdf_sample_grad(const float *pts, const float *boxes, const float *phig, int B, int V, int N, float *g)
{
#pragma omp parallel for schedule(static)
    for (long bi = 0; bi < (long)B * V; ++bi) {
        const int b = (int)(bi / V);
        const float *p = pts + bi * 3, *bx = boxes + b * 4;
        const float *phik = phig + (long)b * N * N * N;
        const float lx = (p[0] - bx[0]) / bx[3], ly = (p[1] - bx[1]) / bx[3], lz = (p[2] - bx[2]) / bx[3];
        const float ix = ((lx + 1.0f) * (float)N - 1.0f) / 2.0f;
        const float iy = ((ly + 1.0f) * (float)N - 1.0f) / 2.0f;
        const float iz = ((lz + 1.0f) * (float)N - 1.0f) / 2.0f;
        const int x0 = (int)floorf(fminf(fmaxf(ix, -4.0f), (float)N + 4.0f));
        const int y0 = (int)floorf(fminf(fmaxf(iy, -4.0f), (float)N + 4.0f));
        const int z0 = (int)floorf(fminf(fmaxf(iz, -4.0f), (float)N + 4.0f));
        float phi[8];
        for (int c = 0; c < 8; ++c) {
            const int xx = x0 + (c & 1), yy = y0 + ((c >> 1) & 1), zz = z0 + (c >> 2);
            phi[c] = (xx >= 0 && xx < N && yy >= 0 && yy < N && zz >= 0 && zz < N) ? phik[((long)zz * N + yy) * N + xx] : 0.f;
        }
        const float x1 = (float)x0 + 1.0f, y1 = (float)y0 + 1.0f, z1 = (float)z0 + 1.0f;
        const float wx[2] = {x1 - ix, ix - (float)x0}, wy[2] = {y1 - iy, iy - (float)y0}, wz[2] = {z1 - iz, iz - (float)z0};
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int c = 0; c < 8; ++c) {
            const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
            const float q = phi[c];
            gx += (dx ? q : -q) * wy[dy] * wz[dz];
            gy += (dy ? q : -q) * wx[dx] * wz[dz];
            gz += (dz ? q : -q) * wx[dx] * wy[dy];
        }
        const float s = (0.5f * (float)N) / bx[3];
        g[bi * 3] = gx * s; g[bi * 3 + 1] = gy * s; g[bi * 3 + 2] = gz * s;
    }
}

/* hm_block_sum of n values taken by `nthreads` threads in strides (csrc/hm_common.h): thread t adds parts[t], parts[t + nthreads],
 * ...; the 64 threads of a wave meet in oc_wave_sum; the wave results are added in wave order. */
float orc_block_sum(const float *parts, int n, int nthreads)
{
    float a[1024];
    for (int t = 0; t < nthreads; ++t) {
        float acc = 0.f;
        for (int i = t; i < n; i += nthreads) acc += parts[i];
        a[t] = acc;
    }
    float tot = 0.f;
    for (int w = 0; w < nthreads / 64; ++w) tot += oc_wave_sum(a + 64 * w);
    return tot;
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Ordinal depth term (reference homan/homan.py:384-419, lossutils.py:133-169, as the method intends: the reference's own call
 * site raises) in the kernels' order: csrc/raster.hip k_ordinal_depth_bwd, k_depth_bwd_faces, k_depth_bwd_gather.
 * ---------------------------------------------------------------------------------------------------------------------- */
static float oc_sigmoid(float xf)      /* csrc/hm_common.h hm_sigmoid */
{
    const double t = -(double)xf;
    const double k = rint(t * 1.44269504088896338700e+00);
    const double r = (t - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    const double z = r * r;
    double p = 4.13813679705723846039e-08;
    p = -1.65339022054652515390e-06 + z * p;
    p = 6.61375632143793436117e-05 + z * p;
    p = -2.77777777770155933842e-03 + z * p;
    p = 1.66666666666666019037e-01 + z * p;
    const double c = r - z * p;
    const double er = 1.0 - ((r * c) / (c - 2.0) - r);
    const double e = ldexp(er, (int)k);
    return (float)(1.0 / (1.0 + e));
}
float orc_sigmoid(float x) { return oc_sigmoid(x); }

/* d (upstream * loss_depth) / d the two pooled depth images.  d0 / d1 (B,S,S) pooled depths, a0 / a1 (B,S,S) uint8: pixel fully
 * covered (pooled alpha == 1), m0 / m1 (B,S,S) uint8 instance masks; layers 0 = object, 1 = hand.  The normalisers are counts
 * (exact): pairs = sum over frames of [layer 0 present] + [layer 1 present] + 2 [both on some pixel]; n01 / n10 = pixels where
 * the annotation puts 0 / 1 in front and the render disagrees.  -> g0, g1 (B,S,S) */
void orc_ordinal_depth_grad(const float *d0, const float *d1, const uint8_t *a0, const uint8_t *a1, const uint8_t *m0,
                            const uint8_t *m1, int B, int S, float upstream, float *g0, float *g1, float *rec)
{
    const long npx = (long)S * S;
    float t0 = 0.f, t1 = 0.f, t3 = 0.f;
    for (int b = 0; b < B; ++b) {
        long c0 = 0, c1 = 0, c01 = 0, n01 = 0, n10 = 0;
        for (long i = b * npx; i < (b + 1) * npx; ++i) {
            c0 += a0[i] != 0; c1 += a1[i] != 0;
            if (a0[i] && a1[i]) {
                ++c01;
                if (m0[i] && !m1[i] && d1[i] < d0[i]) ++n01;
                if (m1[i] && !m0[i] && d0[i] < d1[i]) ++n10;
            }
        }
        t0 += (c0 ? 1.f : 0.f) + (c1 ? 1.f : 0.f) + 2.f * (c01 ? 1.f : 0.f);
        t1 += (float)(unsigned)n01;
        t3 += (float)(unsigned)n10;
    }
    rec[0] = t0; rec[1] = t1; rec[3] = t3; rec[2] = rec[4] = 0.f;
    for (long i = 0; i < B * npx; ++i) {
        float r0 = 0.f, r1 = 0.f;
        if (a0[i] && a1[i]) {
            const float z0 = d0[i], z1 = d1[i];
            const float up = upstream / t0;
            if (m0[i] && !m1[i] && z1 < z0 && t1 > 0.f) {
                const float x = z0 - z1;
                if (x > 0.f && x < 2.f) { const float sg = oc_sigmoid(x); r0 += up * sg / t1; r1 -= up * sg / t1; }
            }
            if (m1[i] && !m0[i] && z0 < z1 && t3 > 0.f) {
                const float x = z1 - z0;
                if (x > 0.f && x < 2.f) { const float sg = oc_sigmoid(x); r1 += up * sg / t3; r0 -= up * sg / t3; }
            }
        }
        g0[i] = r0; g1[i] = r1;
    }
}

static inline float oc_topix2(float v, int is)
{
    float a = v * (float)is;
    a = a + (float)is;
    a = a - 1.0f;
    return 0.5f * a;
}
static inline int oc_backside2(const float *f) { return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]); }

/* Backward of the pooled depth image w.r.t. the NDC face vertices (NMR backward_depth_map), per (frame, face, winding) in the
 * order of k_depth_bwd_faces: the face's sample box (k_setup_faces: pixel-space extent, 0.01 px of slack, clipped) is walked
 * row-major by 64 lanes in strides, each lane adding its samples in turn, the lanes meeting in oc_wave_sum.
 * faces9 (B,F,9): NDC vertices of the mesh faces; idx_map (B,is,is) owner per sample (face, or F + face for the reversed
 * winding, -1 empty), is = 2 S; gpd (B,S,S): gradient on the pooled depth.  -> gf9 (B,F,2,9) in winding order */
void orc_depth_bwd_faces(const float *faces9, const int32_t *idx_map, const float *gpd, int B, int F, int S, float *gf9)
{
    const int is = 2 * S;
#pragma omp parallel for schedule(dynamic, 64)
    for (long bf = 0; bf < (long)B * F; ++bf) {
        const int b = (int)(bf / F), fi = (int)(bf % F);
        const float *src = faces9 + bf * 9;
        const int32_t *idx = idx_map + (long)b * is * is;
        const float *g = gpd + (long)b * S * S;
        float rv[9];
        for (int k = 0; k < 3; ++k) { rv[3 * k] = src[3 * (2 - k)]; rv[3 * k + 1] = src[3 * (2 - k) + 1]; rv[3 * k + 2] = src[3 * (2 - k) + 2]; }
        unsigned mask = (oc_backside2(src) ? 0u : 1u) | (oc_backside2(rv) ? 0u : 2u);
        float bpx[3], bpy[3];
        for (int k = 0; k < 3; ++k) { bpx[k] = oc_topix2(src[3 * k], is); bpy[k] = oc_topix2(src[3 * k + 1], is); }
        const float xmin = fminf(bpx[0], fminf(bpx[1], bpx[2])), xmax = fmaxf(bpx[0], fmaxf(bpx[1], bpx[2]));
        const float ymin = fminf(bpy[0], fminf(bpy[1], bpy[2])), ymax = fmaxf(bpy[0], fmaxf(bpy[1], bpy[2]));
        if (!(xmax >= -2.0f && ymax >= -2.0f && xmin <= is + 1.0f && ymin <= is + 1.0f)) mask = 0;
        int x0 = (int)ceilf(fmaxf(xmin, -2.0f) - 0.01f), x1 = (int)floorf(fminf(xmax, is + 1.0f) + 0.01f);
        int y0 = (int)ceilf(fmaxf(ymin, -2.0f) - 0.01f), y1 = (int)floorf(fminf(ymax, is + 1.0f) + 0.01f);
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 > is - 1) x1 = is - 1;
        if (y1 > is - 1) y1 = is - 1;
        if (x1 < x0 || y1 < y0) mask = 0;
        for (int var = 0; var < 2; ++var) {
            float *out = gf9 + (bf * 2 + var) * 9;
            const int fn = fi + var * F;
            for (int k = 0; k < 9; ++k) out[k] = 0.f;
            if (!((mask >> var) & 1u)) continue;
            const int bw = x1 - x0 + 1, n = bw * (y1 - y0 + 1);
            int owns = 0;
            for (int e = 0; e < n && !owns; ++e) owns = idx[(long)(y0 + e / bw) * is + x0 + e % bw] == fn;
            if (!owns) continue;
            float f[9];
            for (int k = 0; k < 3; ++k) {
                const int sv = var ? 2 - k : k;
                f[3 * k] = src[3 * sv]; f[3 * k + 1] = src[3 * sv + 1]; f[3 * k + 2] = src[3 * sv + 2];
            }
            float p[3][2];
            for (int k = 0; k < 3; ++k) { p[k][0] = oc_topix2(f[3 * k], is); p[k][1] = oc_topix2(f[3 * k + 1], is); }
            float inv[9] = {
                p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
            const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
            for (int k = 0; k < 9; ++k) inv[k] = inv[k] / den;
            const float rz0 = 1.0f / f[2], rz1 = 1.0f / f[5], rz2 = 1.0f / f[8];
            float L0[64], L1[64], L2[64];
            for (int lane = 0; lane < 64; ++lane) {
                float A0 = 0.f, A1 = 0.f, A2 = 0.f;
                for (int e = lane; e < n; e += 64) {
                    const int xi = x0 + e % bw, yi = y0 + e / bw;
                    if (idx[(long)yi * is + xi] != fn) continue;
                    float wgt[3], ws = 0.f;
                    for (int k = 0; k < 3; ++k) {
                        float t = inv[3 * k] * (float)xi;
                        t = t + inv[3 * k + 1] * (float)yi;
                        t = t + inv[3 * k + 2];
                        t = fminf(fmaxf(t, 0.0f), 1.0f);
                        wgt[k] = t;
                        ws += t;
                    }
                    float sum = wgt[0] * rz0;
                    sum = sum + wgt[1] * rz1;
                    sum = sum + wgt[2] * rz2;
                    const float zp = ws / sum;
                    const float a = 0.25f * g[(long)((is - 1 - yi) >> 1) * S + (xi >> 1)] * zp * zp;
                    A0 += a * (wgt[0] / ws); A1 += a * (wgt[1] / ws); A2 += a * (wgt[2] / ws);
                }
                L0[lane] = A0; L1[lane] = A1; L2[lane] = A2;
            }
            const float A[3] = {oc_wave_sum(L0), oc_wave_sum(L1), oc_wave_sum(L2)}, rz[3] = {rz0, rz1, rz2};
            const float tmp0 = -(inv[0] * rz0 + inv[3] * rz1 + inv[6] * rz2);
            const float tmp1 = -(inv[1] * rz0 + inv[4] * rz1 + inv[7] * rz2);
            const float half_is = 0.5f * (float)is;
            for (int k = 0; k < 3; ++k) {
                out[3 * k] = -A[k] * tmp0 * half_is;
                out[3 * k + 1] = -A[k] * tmp1 * half_is;
                out[3 * k + 2] = A[k] * rz[k] * rz[k];
            }
        }
    }
}

/* vertex gather of gf9 (winding order -> mesh corners, the adjacency lists in ascending (face, corner) order) + backward of
 * the projection (z passes straight through): k_depth_bwd_gather.  verts (B,V,3) camera space, K (B,3,3) -> grad (B,V,3) */
void orc_depth_bwd_gather(const float *gf9, const int32_t *adj_off, const int32_t *adj_items, const float *verts, const float *K,
                          int B, int V, int F, float orig_size, float *grad_verts)
{
    for (long i = 0; i < (long)B * V; ++i) {
        const int b = (int)(i / V), v = (int)(i % V);
        float gu = 0.f, gv = 0.f, gz = 0.f;
        for (int a = adj_off[v]; a < adj_off[v + 1]; ++a) {
            const int item = adj_items[a], fi = item / 3, k = item % 3;
            const float *pf = gf9 + ((long)b * F + fi) * 18;
            gu += pf[3 * k] + pf[9 + 3 * (2 - k)];
            gv += pf[3 * k + 1] + pf[9 + 3 * (2 - k) + 1];
            gz += pf[3 * k + 2] + pf[9 + 3 * (2 - k) + 2];
        }
        const float *k = K + b * 9;
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        const float zz = z + 1e-9f;
        const float du0 = gu * (2.0f / orig_size), dv0 = -gv * (2.0f / orig_size);
        const float dxn = k[0] * du0 + k[3] * dv0;
        const float dyn = k[1] * du0 + k[4] * dv0;
        grad_verts[3 * i] = dxn / zz;
        grad_verts[3 * i + 1] = dyn / zz;
        grad_verts[3 * i + 2] = -(dxn * x + dyn * y) / (zz * zz) + gz;
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Pose initialisation: off-screen penalty (reference homan/pose_optimization.py:112-135) in the order of csrc/losses.hip
 * k_offscreen - hinges on the six clipping planes per vertex, element-wise gradient; the value of a candidate = weight * the
 * sum over its vertices taken by `nthreads` threads in strides, waves through oc_wave_sum, wave results in wave order.
 * verts (n,V,3), K: ONE 3x3 camera -> out (n), grad (n,V,3) */
void orc_offscreen(const float *verts, const float *K, int n, int V, float zfar, float weight, int nthreads, float *out, float *grad)
{
    const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5];
#pragma omp parallel for schedule(static)
    for (int c = 0; c < n; ++c) {
        float acc[1024];
        for (int t = 0; t < nthreads; ++t) acc[t] = 0.f;
        for (int v = 0; v < V; ++v) {
            const long o = ((long)c * V + v) * 3;
            const float x = verts[o], y = verts[o + 1], z = verts[o + 2];
            const float zz = z + 1e-9f;
            const float xn = x / zz, yn = y / zz;
            float u = k00 * xn + k01 * yn;
            u = u + k02;
            float w = k10 * xn + k11 * yn;
            w = 1.0f - (w + k12);
            const float nu = 2.0f * (u - 0.5f), nv = 2.0f * (w - 0.5f);
            float val = fmaxf(nu - 1.0f, 0.f) + fmaxf(nv - 1.0f, 0.f);
            val += fmaxf(-1.0f - nu, 0.f) + fmaxf(-1.0f - nv, 0.f);
            val += fmaxf(-z, 0.f);
            val += fmaxf(z - zfar, 0.f);
            acc[v % nthreads] += val;          /* thread t = v mod nthreads meets its vertices in ascending order */
            const float gu = (nu - 1.0f > 0.f ? 1.f : 0.f) - (-1.0f - nu > 0.f ? 1.f : 0.f);
            const float gv = (nv - 1.0f > 0.f ? 1.f : 0.f) - (-1.0f - nv > 0.f ? 1.f : 0.f);
            const float gz = (z - zfar > 0.f ? 1.f : 0.f) - (-z > 0.f ? 1.f : 0.f);
            const float du = 2.0f * gu, dw = -2.0f * gv;
            const float dxn = k00 * du + k10 * dw, dyn = k01 * du + k11 * dw;
            grad[o] = weight * (dxn / zz);
            grad[o + 1] = weight * (dyn / zz);
            grad[o + 2] = weight * (gz - (dxn * x + dyn * y) / (zz * zz));
        }
        float tot = 0.f;
        for (int w = 0; w < nthreads / 64; ++w) tot += oc_wave_sum(acc + 64 * w);
        out[c] = weight * tot;
    }
}
