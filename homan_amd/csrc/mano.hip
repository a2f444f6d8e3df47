// mano.hip -- fused MANO linear-blend-skinning forward + backward for CDNA4.
//
// Replaces, on the reference's hot path, ManoModel.forward_pca (reference homan/manomodel.py:84-151:
// hand_pose = pca[:, :16] @ components + hand_mean; layer(betas, global_orient, hand_pose, transl=0))
// and the third-party `mano` layer it calls (smplx-style LBS: shape blend shapes, joint regression,
// Rodrigues with angle=|r+1e-8|, pose blend shapes on (R[1:]-I), kinematic chain, skinning), plus the
// "+ mano_trans" of homan/homan.py:356.  ~40 tiny torch kernels per call become one launch forward and two
// launches backward.
//
// Layout: M (145 x 2334) = [posedirs (135 rows) ; shapedirs^T (10 rows)], row-major in vertex-coordinate
// index 3v+c, so a wave reading one row for 64 consecutive vertices is coalesced; the joint regressor is
// folded on the host into J_template (16x3) + J_shapedirs (16x3x10).
#include "hm_common.h"
#include <string.h>

#define MANO_V 778
#define MANO_J 16
#define MANO_NF 145   // 135 pose-blend features + 10 betas
#define MANO_CHUNK 256
#define MANO_NCHUNK 4  // ceil(778/256)
#define MANO_PART 352  // per-(frame,chunk) partials: 192 dA + 145 dfeat + 3 dtrans + 12 of the fused rigid backward (dR, dt)
#define MANO_NCH64 13  // ceil(778/64) vertex chunks of 64
// forward state kept for the backward (per frame): the chain state (struct ManoShared, dword copy) + the posed vertices
#define MANO_STATE_SH 832
#define MANO_STATE_DW (MANO_STATE_SH + 3 * MANO_V + 2)

struct ManoModelDev {
    const float* v_template;   // (778,3)
    const float* M;            // (145, 2334)
    const float* J_template;   // (16,3)
    const float* J_shapedirs;  // (16,3,10)
    const float* weights;      // (778,16)
    const float* comps;        // (16,45)
    const float* hand_mean;    // (45)
    const int* parents;        // (16)
};

__device__ __forceinline__ void rodrigues(const float* r, float* R)
{
    const float e0 = r[0] + 1e-8f, e1 = r[1] + 1e-8f, e2 = r[2] + 1e-8f;
    const float a = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
    const float nx = r[0] / a, ny = r[1] / a, nz = r[2] / a;
    float s, cs_;
    hm_sincos(a, &s, &cs_);          // (a defined function of a: the oracle evaluates the same operations)
    const float c1 = 1.0f - cs_;
    // K = [[0,-nz,ny],[nz,0,-nx],[-ny,nx,0]] ; R = I + s K + (1-c) K K
    const float K[9] = {0.f, -nz, ny, nz, 0.f, -nx, -ny, nx, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float kk = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
            R[3 * i + j] = (i == j ? 1.0f : 0.0f) + s * K[3 * i + j] + c1 * kk;
        }
}

// dL/dR -> dL/dr
__device__ __forceinline__ void rodrigues_backward(const float* r, const float* dR, float* dr)
{
    const float e[3] = {r[0] + 1e-8f, r[1] + 1e-8f, r[2] + 1e-8f};
    const float a = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    const float n[3] = {r[0] / a, r[1] / a, r[2] / a};
    float s, c;
    hm_sincos(a, &s, &c);
    const float c1 = 1.0f - c;
    const float K[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KK[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    float dRK = 0.f, dRKK = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { dRK += dR[k] * K[k]; dRKK += dR[k] * KK[k]; }
    float da = c * dRK + s * dRKK;
    // dK = s dR + (1-c) (dR K^T + K^T dR)
    float dK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float t1 = dR[3 * i] * K[3 * j] + dR[3 * i + 1] * K[3 * j + 1] + dR[3 * i + 2] * K[3 * j + 2];   // dR K^T
            float t2 = K[i] * dR[j] + K[3 + i] * dR[3 + j] + K[6 + i] * dR[6 + j];                            // K^T dR
            dK[3 * i + j] = s * dR[3 * i + j] + c1 * (t1 + t2);
        }
    const float dn[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    // n = r / a
    da += -(dn[0] * r[0] + dn[1] * r[1] + dn[2] * r[2]) / (a * a);
#pragma unroll
    for (int i = 0; i < 3; ++i) dr[i] = dn[i] / a + da * e[i] / a;
}

struct ManoShared {
    float pose[48];
    float Rl[MANO_J][9];     // local rotations
    float J[MANO_J][3];      // rest joints
    float Rw[MANO_J][9];     // world rotations
    float tw[MANO_J][3];     // world translations (= posed joints)
    float A[MANO_J][12];     // skinning transforms [R | t] with the rest pose removed
    float feat[MANO_NF];     // pose feature (135) + betas (10)
    int parents[MANO_J];     // kinematic tree, staged once (chasing it through global memory is a chain of dependent loads)
    int depth[MANO_J];
};

// pose / Rodrigues / joints / chain, shared by forward and both backward kernels.  `t` = index of the thread among the
// >= 64 that prepare frame b into `sh` (the barriers are the workgroup's: every wave of a workgroup calls this together,
// each group of threads with a frame and a state of its own - or all of them with t = threadIdx.x and one frame).
__device__ __forceinline__ void mano_prepare(const ManoModelDev& m, const float* pca, int pca_stride, const float* rot,
                                             const float* betas, int b, ManoShared& sh, int t)
{
    if (t < 48) {
        float v;
        if (t < 3) v = rot[b * 3 + t];
        else {
            const int k = t - 3;
            v = 0.f;
            for (int i = 0; i < 16; ++i) v += pca[(long)b * pca_stride + i] * m.comps[i * 45 + k];
            v += m.hand_mean[k];
        }
        sh.pose[t] = v;
    }
    if (t < 10) sh.feat[135 + t] = betas[b * 10 + t];
    if (t < MANO_J) sh.parents[t] = m.parents[t];
    __syncthreads();
    if (t < MANO_J) {
        int d = 0;
        for (int q = sh.parents[t]; q >= 0; q = sh.parents[q]) ++d;
        sh.depth[t] = d;
        float R[9];
        rodrigues(&sh.pose[3 * t], R);
#pragma unroll
        for (int k = 0; k < 9; ++k) sh.Rl[t][k] = R[k];
        if (t >= 1)
#pragma unroll
            for (int k = 0; k < 9; ++k) sh.feat[9 * (t - 1) + k] = R[k] - ((k % 4 == 0) ? 1.0f : 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = m.J_template[t * 3 + c];
            for (int l = 0; l < 10; ++l) v += betas[b * 10 + l] * m.J_shapedirs[(t * 3 + c) * 10 + l];
            sh.J[t][c] = v;
        }
    }
    __syncthreads();
    // kinematic chain, level by level (joint t waits until its parent's level is done; MANO depth is 3)
    int depth = 0, par = -1, maxd = 0;
    if (t < MANO_J) { par = sh.parents[t]; depth = sh.depth[t]; }
#pragma unroll
    for (int j = 0; j < MANO_J; ++j) maxd = max(maxd, sh.depth[j]);     // depth of the tree, the same in every thread
    for (int level = 0; level < MANO_J; ++level) {
        if (t < MANO_J && depth == level) {
            const int j = t, p = par;
            if (p < 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) sh.Rw[j][k] = sh.Rl[j][k];
#pragma unroll
                for (int c = 0; c < 3; ++c) sh.tw[j][c] = sh.J[j][c];
            } else {
                const float rel[3] = {sh.J[j][0] - sh.J[p][0], sh.J[j][1] - sh.J[p][1], sh.J[j][2] - sh.J[p][2]};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        sh.Rw[j][3 * i + k] = sh.Rw[p][3 * i] * sh.Rl[j][k] + sh.Rw[p][3 * i + 1] * sh.Rl[j][3 + k] +
                                              sh.Rw[p][3 * i + 2] * sh.Rl[j][6 + k];
                    sh.tw[j][i] = sh.Rw[p][3 * i] * rel[0] + sh.Rw[p][3 * i + 1] * rel[1] + sh.Rw[p][3 * i + 2] * rel[2] +
                                  sh.tw[p][i];
                }
            }
        }
        __syncthreads();
        if (level >= maxd) break;          // block-uniform
    }
    if (t < MANO_J) {
        const int j = t;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sh.A[j][4 * i] = sh.Rw[j][3 * i];
            sh.A[j][4 * i + 1] = sh.Rw[j][3 * i + 1];
            sh.A[j][4 * i + 2] = sh.Rw[j][3 * i + 2];
            sh.A[j][4 * i + 3] = sh.tw[j][i] - (sh.Rw[j][3 * i] * sh.J[j][0] + sh.Rw[j][3 * i + 1] * sh.J[j][1] +
                                                sh.Rw[j][3 * i + 2] * sh.J[j][2]);
        }
    }
    __syncthreads();
}

// ---- vertex stage.  Workgroup = 64 vertices x 4 wavefronts: wave w accumulates rows [w*37, w*37+37) of the blend
// matrix for the 64 vertices (three coalesced 4-byte loads per row, rows independent -> loads pipeline), the four
// partial sums meet in LDS.
#define MANO_VCH 64
#define MANO_ROWS_PER_WAVE 37
__device__ __forceinline__ void mano_posed_chunk(const ManoModelDev& m, const ManoShared& sh, int v0, float (*s_part)[MANO_VCH][3],
                                                 float (*s_vp)[3])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v = min(v0 + lane, MANO_V - 1);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int k0 = wv * MANO_ROWS_PER_WAVE, k1 = min(MANO_NF, k0 + MANO_ROWS_PER_WAVE);
    const float* base = m.M + 3 * v;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const float f = sh.feat[k];
        const float* row = base + (long)k * (3 * MANO_V);
        a0 += f * row[0];
        a1 += f * row[1];
        a2 += f * row[2];
    }
    s_part[wv][lane][0] = a0; s_part[wv][lane][1] = a1; s_part[wv][lane][2] = a2;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            s_vp[lane][c] = m.v_template[3 * v + c] + ((s_part[0][lane][c] + s_part[1][lane][c]) + (s_part[2][lane][c] + s_part[3][lane][c]));
    }
    __syncthreads();
}

__device__ __forceinline__ void mano_skin_transform(const ManoModelDev& m, const ManoShared& sh, int v, float* T)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
    const float4* wrow = reinterpret_cast<const float4*>(m.weights + v * MANO_J);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w4 = wrow[q];
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * q + u;
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] += ws[u] * sh.A[j][k];
        }
    }
}

// grid (13, ceil(B / 4)).  verts (B,778,3) = LBS + trans ; joints (B,16,3) optional (posed joints + trans).
// A workgroup = one chunk of 64 vertices for FOUR consecutive frames, wave f owning frame f: the four chains are prepared
// side by side (one per wave), then every wave streams ITS 37 rows of the blend matrix once and accumulates them for all
// four frames - the 1.35 MB matrix is read from L2 once per four frames instead of once per frame (a 240-frame batch moved
// 324 MB through L2 per launch) - and wave f finishes frame f: partial sums, skinning, rigid transform, state for the backward.
// Per frame the arithmetic is unchanged (same rows per partial, same order), so results do not depend on how frames are grouped.
#define MANO_FPW 4     // frames per workgroup = waves per workgroup
__global__ __launch_bounds__(256) void k_mano_fwd(ManoModelDev m, const float* __restrict__ pca, int pca_stride,
                                                   const float* __restrict__ rot, const float* __restrict__ betas,
                                                   const float* __restrict__ trans, int B, float* __restrict__ verts,
                                                   float* __restrict__ joints, const float* __restrict__ rigid_rot6d,
                                                   const float* __restrict__ rigid_trans,
                                                   const float* __restrict__ rigid_scale, float* __restrict__ verts_world,
                                                   float* __restrict__ state, int clip_len, int row0, int row_stride)
{
    __shared__ ManoShared shs[MANO_FPW];
    __shared__ float s_part[4][MANO_FPW][MANO_VCH][3];
    __shared__ float s_vp[MANO_FPW][MANO_VCH][3];
    __shared__ float s_R[MANO_FPW][9];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b_raw = blockIdx.y * MANO_FPW + wv;
    const bool live = b_raw < B;                 // (a frame past the end repeats the last one and stores nothing)
    // frame of this launch -> row of the caller's arrays: row0 + frame * row_stride (two hands per frame are interleaved
    // frame-major, reference homan.py:62-63: hand i of every frame = rows i, i + 2, ... through ITS side's model; one hand:
    // row == frame).  Every per-row array - parameters, vertices, state - is addressed by the row; clip_len counts rows.
    const int b = (min(b_raw, B - 1)) * row_stride + row0;
    ManoShared& sh = shs[wv];
    if (verts_world && lane == 48) rot6d_to_mat(rigid_rot6d + b * 6, s_R[wv]);      // published by mano_prepare's barriers
    mano_prepare(m, pca, pca_stride, rot, betas, b, sh, lane);
    const float tr[3] = {trans ? trans[b * 3] : 0.f, trans ? trans[b * 3 + 1] : 0.f, trans ? trans[b * 3 + 2] : 0.f};
    if (joints && live && blockIdx.x == 0 && lane < MANO_J * 3)
        joints[b * MANO_J * 3 + lane] = sh.tw[lane / 3][lane % 3] + tr[lane % 3];
    const int v0 = blockIdx.x * MANO_VCH;
    const int v = min(v0 + lane, MANO_V - 1);
    {
        float a[MANO_FPW][3];
#pragma unroll
        for (int f = 0; f < MANO_FPW; ++f) a[f][0] = a[f][1] = a[f][2] = 0.f;
        const int k0 = wv * MANO_ROWS_PER_WAVE, k1 = min(MANO_NF, k0 + MANO_ROWS_PER_WAVE);
        const float* base = m.M + 3 * v;
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const float* row = base + (long)k * (3 * MANO_V);
            const float r0 = row[0], r1 = row[1], r2 = row[2];
#pragma unroll
            for (int f = 0; f < MANO_FPW; ++f) {
                const float ft = shs[f].feat[k];
                a[f][0] += ft * r0;
                a[f][1] += ft * r1;
                a[f][2] += ft * r2;
            }
        }
#pragma unroll
        for (int f = 0; f < MANO_FPW; ++f) {
            s_part[wv][f][lane][0] = a[f][0]; s_part[wv][f][lane][1] = a[f][1]; s_part[wv][f][lane][2] = a[f][2];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c)
        s_vp[wv][lane][c] = m.v_template[3 * v + c] + ((s_part[0][wv][lane][c] + s_part[1][wv][lane][c]) +
                                                       (s_part[2][wv][lane][c] + s_part[3][wv][lane][c]));
    __builtin_amdgcn_wave_barrier();
    if (!live) return;
    if (state) {        // chain state once per frame, posed vertices per chunk: the backward reloads instead of recomputing
        float* st = state + (long)b * MANO_STATE_DW;
        if (blockIdx.x == 0)
            for (int i = lane; i < (int)(sizeof(ManoShared) / 4); i += 64) st[i] = reinterpret_cast<const float*>(&sh)[i];
        const int nv3 = 3 * min(MANO_VCH, MANO_V - v0);
        for (int i = lane; i < nv3; i += 64) st[MANO_STATE_SH + 3 * v0 + i] = (&s_vp[wv][0][0])[i];
    }
    if (v0 + lane >= MANO_V) return;
    float T[12];
    mano_skin_transform(m, sh, v, T);
    const float* vp = s_vp[wv][lane];
    float* o = verts + ((long)b * MANO_V + v) * 3;
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { p[i] = T[4 * i] * vp[0] + T[4 * i + 1] * vp[1] + T[4 * i + 2] * vp[2] + T[4 * i + 3] + tr[i]; o[i] = p[i]; }
    if (verts_world) {       // the rigid hand transform of homan.py:341-382, same arithmetic as k_rigid_fwd (no abs on the scale)
        const float s = rigid_scale[b / clip_len];      // one hand scale per clip
        const float x = s * p[0], y = s * p[1], z = s * p[2];
        const float* t = rigid_trans + b * 3;
        const float* R = s_R[wv];
        float* ow = verts_world + ((long)b * MANO_V + v) * 3;
        ow[0] = x * R[0] + y * R[3] + z * R[6] + t[0];
        ow[1] = x * R[1] + y * R[4] + z * R[7] + t[1];
        ow[2] = x * R[2] + y * R[5] + z * R[8] + t[2];
    }
}

static_assert(sizeof(ManoShared) / 4 <= MANO_STATE_SH, "MANO_STATE_SH too small");

// ---- backward, second half (one workgroup per frame: the one that finishes the frame's last vertex chunk): reduce the
// chunk partials, chain / Rodrigues / PCA backward (level-parallel).  `sh` holds the frame's chain state already.
struct ManoBwd2Shared {
    float tot[MANO_PART];
    float dRw[MANO_J][9], dtw[MANO_J][3], dJ[MANO_J][3], dRl[MANO_J][9], dpose[48];
    float cR[MANO_J][9], ct[MANO_J][3], cJ[MANO_J][3];     // child -> parent contributions
};
__device__ __forceinline__ void mano_bwd2_body(const ManoModelDev& m, const ManoShared& sh, ManoBwd2Shared& w,
                                               const float* __restrict__ partials, int nchunk, int lb, int b, int pca_dim,
                                               const float* __restrict__ g_pca_extra, float w_extra,
                                               float* __restrict__ g_pca, float* __restrict__ g_rot,
                                               float* __restrict__ g_betas, float* __restrict__ g_trans)
{
    const int t = threadIdx.x, nt = blockDim.x;
    for (int k = t; k < MANO_PART; k += nt) {
        float v[MANO_NCH64];
#pragma unroll
        for (int c = 0; c < MANO_NCH64; ++c)        // all requests first, then the (fixed order) sum
            v[c] = c < nchunk ? hm_partial_load(partials + ((long)lb * nchunk + c) * MANO_PART + k) : 0.f;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < MANO_NCH64; ++c) a += v[c];
        w.tot[k] = a;
    }
    int depth = 0, par = -1, maxd = 0;
    if (t < MANO_J) { par = sh.parents[t]; depth = sh.depth[t]; }
#pragma unroll
    for (int j = 0; j < MANO_J; ++j) maxd = max(maxd, sh.depth[j]);
    __syncthreads();
    if (t < MANO_J) {
        const int j = t;
        const float* dA = &w.tot[j * 12];
        // A = [Rw | tw - Rw J]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) w.dRw[j][3 * i + k] = dA[4 * i + k] - dA[4 * i + 3] * sh.J[j][k];
            w.dtw[j][i] = dA[4 * i + 3];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            w.dJ[j][k] = -(sh.Rw[j][k] * dA[3] + sh.Rw[j][3 + k] * dA[7] + sh.Rw[j][6 + k] * dA[11]);
    }
    __syncthreads();
    // leaves first: joints of one level push their contribution to per-child slots, then every parent collects
    for (int level = maxd; level >= 1; --level) {
        if (t < MANO_J && depth == level) {
            const int j = t, p = par;
            const float rel[3] = {sh.J[j][0] - sh.J[p][0], sh.J[j][1] - sh.J[p][1], sh.J[j][2] - sh.J[p][2]};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k)       // tw_j = Rw_p rel + tw_p ; Rw_j = Rw_p Rl_j
                    w.cR[j][3 * i + k] = w.dtw[j][i] * rel[k] + w.dRw[j][3 * i] * sh.Rl[j][3 * k] +
                                         w.dRw[j][3 * i + 1] * sh.Rl[j][3 * k + 1] + w.dRw[j][3 * i + 2] * sh.Rl[j][3 * k + 2];
                w.ct[j][i] = w.dtw[j][i];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = sh.Rw[p][k] * w.dtw[j][0] + sh.Rw[p][3 + k] * w.dtw[j][1] + sh.Rw[p][6 + k] * w.dtw[j][2];
                w.dJ[j][k] += d;
                w.cJ[j][k] = -d;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    w.dRl[j][3 * i + k] = sh.Rw[p][i] * w.dRw[j][k] + sh.Rw[p][3 + i] * w.dRw[j][3 + k] + sh.Rw[p][6 + i] * w.dRw[j][6 + k];
        }
        __syncthreads();
        if (t < MANO_J && depth == level - 1) {
            for (int j = 0; j < MANO_J; ++j)
                if (sh.parents[j] == t) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) w.dRw[t][k] += w.cR[j][k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { w.dtw[t][k] += w.ct[j][k]; w.dJ[t][k] += w.cJ[j][k]; }
                }
        }
        __syncthreads();
    }
    if (t < MANO_J && par < 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) w.dRl[t][k] = w.dRw[t][k];
#pragma unroll
        for (int c = 0; c < 3; ++c) w.dJ[t][c] += w.dtw[t][c];
    }
    __syncthreads();
    if (t < MANO_J) {
        float dR[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[k] = w.dRl[t][k] + (t >= 1 ? w.tot[192 + 9 * (t - 1) + k] : 0.f);
        float dr[3];
        rodrigues_backward(&sh.pose[3 * t], dR, dr);
#pragma unroll
        for (int c = 0; c < 3; ++c) w.dpose[3 * t + c] = dr[c];
    }
    __syncthreads();
    if (t < 3) {
        g_rot[b * 3 + t] = w.dpose[t];
        g_trans[b * 3 + t] = w.tot[337 + t];
    }
    for (int i = t; i < pca_dim; i += nt) {
        float a = 0.f;
        if (i < 16) {
            float cv[45];
#pragma unroll
            for (int k = 0; k < 45; ++k) cv[k] = m.comps[i * 45 + k];            // 45 independent loads in flight
#pragma unroll
            for (int k = 0; k < 45; ++k) a += cv[k] * w.dpose[3 + k];
        }
        if (g_pca_extra) a += w_extra * g_pca_extra[(long)b * pca_dim + i];      // e.g. the PCA prior's unit gradient
        g_pca[(long)b * pca_dim + i] = a;
    }
    if (t < 10) {
        float a = w.tot[192 + 135 + t];
        float jv[MANO_J * 3];
#pragma unroll
        for (int q = 0; q < MANO_J * 3; ++q) jv[q] = m.J_shapedirs[q * 10 + t];
#pragma unroll
        for (int q = 0; q < MANO_J * 3; ++q) a += jv[q] * w.dJ[q / 3][q % 3];
        g_betas[b * 10 + t] = a;
    }
}

#ifdef MANO_PHASES
static __device__ unsigned long long g_mano_ph[16];
#define MPH_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_mano_ph[k], t_ - mph_t); mph_t = t_; } } while (0)
#else
#define MPH_MARK(k)
#endif
// Optional (mesh != NULL): the backward of the hand's RIGID transform (k_rigid_bwd, geometry.hip) inside this launch.  The
// gradient w.r.t. the model-space vertex that `gout` would hold is formed on the fly from the per-vertex terms on the
// world-space vertex (same expressions as k_rigid_bwd), the chunk's twelve sums of dR / dt ride the chunk record, and the
// frame's finishing workgroup turns them into d rot6d / d translation: one launch less on the hand-side chain.
struct ManoRigid {
    const float* mesh;            // (rows, 778, 3): the forward's model-space vertices
    const float* rot6d;           // (rows, 3, 2)
    const float* scale;           // one per clip of clip_len rows
    int clip_len;
    RigidTerms terms;             // weighted gradient terms on the world-space vertices (reach mesh, R, t)
    const float* g_rigid;         // optional per-vertex term that reaches R, t only
    const float* g_frame;         // optional per-row vector (frame_stride floats apart, times frame_scale) that reaches R, t only
    int frame_stride;
    float frame_scale;
    float* g_rot6d; float* g_trans;
};
// backward: grid (13, B).  First half per (vertex chunk, frame) -> partials (B, 13, MANO_PART); the workgroup that finishes
// the last chunk of a frame (per-frame ticket) runs the second half for that frame -- one launch, and the chain state is
// reloaded from the forward (`state`) instead of being recomputed when the caller kept it.
template <bool RIGID>
__global__ __launch_bounds__(256) void k_mano_bwd(ManoModelDev m, const float* __restrict__ pca, int pca_stride,
                                                   const float* __restrict__ rot, const float* __restrict__ betas,
                                                   const float* __restrict__ gout, int B, const float* __restrict__ state,
                                                   float* __restrict__ partials, unsigned int* __restrict__ frame_cnt,
                                                   int pca_dim, const float* __restrict__ g_pca_extra, float w_extra,
                                                   float* __restrict__ g_pca, float* __restrict__ g_rot,
                                                   float* __restrict__ g_betas, float* __restrict__ g_trans, int row0,
                                                   int row_stride, ManoRigid mr)
{
    HM_HAND_KERNEL();
    __shared__ ManoShared sh;
    __shared__ float s_part[4][MANO_VCH][3];
    __shared__ float s_vp[MANO_VCH][3];
    __shared__ float s_g[MANO_VCH][3];
    __shared__ float s_dvp[MANO_VCH * 3];
    __shared__ float s_w[MANO_VCH][MANO_J + 1];
    __shared__ float red[16];
    __shared__ ManoBwd2Shared w2;
    __shared__ int s_flag;
    // lb: frame of this launch (partials, tickets); b: its row in the caller's arrays (see k_mano_fwd)
    const int lb = blockIdx.y, b = lb * row_stride + row0, t = threadIdx.x;
    const int v0 = blockIdx.x * MANO_VCH;
    const int nv = min(MANO_VCH, MANO_V - v0);
#ifdef MANO_PHASES
    unsigned long long mph_t = wall_clock64();
    if (t == 0) atomicAdd(&g_mano_ph[15], 1ull);
#endif
    if (state) {
        const float* st = state + (long)b * MANO_STATE_DW;
        for (int i = t; i < (int)(sizeof(ManoShared) / 4); i += blockDim.x) reinterpret_cast<float*>(&sh)[i] = st[i];
        if (t < 3 * nv) (&s_vp[0][0])[t] = st[MANO_STATE_SH + 3 * v0 + t];
        __syncthreads();
    } else {
        mano_prepare(m, pca, pca_stride, rot, betas, b, sh, threadIdx.x);
        mano_posed_chunk(m, sh, v0, s_part, s_vp);
    }
    MPH_MARK(0);
    float g[3] = {0.f, 0.f, 0.f};
    float racc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) racc[k] = 0.f;
    if (t < nv) {
        const int v = v0 + t;
        if (RIGID) {
            // rigid backward of this vertex (the arithmetic of k_rigid_bwd): gf reaches the mesh, gt = gf + the rigid-only terms
            const long o = ((long)b * MANO_V + v) * 3;
            float R[9];
            rot6d_to_mat(mr.rot6d + (long)b * 6, R);
            const float s = mr.scale[b / mr.clip_len];
            const float mv[3] = {mr.mesh[o], mr.mesh[o + 1], mr.mesh[o + 2]};
            float gf[3] = {0.f, 0.f, 0.f}, gt[3];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (mr.terms.p[k]) {
                    gf[0] += mr.terms.w[k] * mr.terms.p[k][o]; gf[1] += mr.terms.w[k] * mr.terms.p[k][o + 1];
                    gf[2] += mr.terms.w[k] * mr.terms.p[k][o + 2];
                }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                gt[c] = gf[c] + (mr.g_frame ? mr.frame_scale * mr.g_frame[(long)b * mr.frame_stride + c] : 0.f);
            if (mr.g_rigid) { gt[0] += mr.g_rigid[o]; gt[1] += mr.g_rigid[o + 1]; gt[2] += mr.g_rigid[o + 2]; }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) racc[3 * i + j] = (s * mv[i]) * gt[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) racc[9 + j] = gt[j];
            // d(s*m)_i = sum_j R[i][j] gf_j ; d m = s * that
            g[0] = s * (R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2]);
            g[1] = s * (R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2]);
            g[2] = s * (R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]);
        } else {
            const float* gp = gout + ((long)b * MANO_V + v) * 3;
            g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
        }
    }
    float* out = partials + ((long)lb * gridDim.x + blockIdx.x) * MANO_PART;
    if (RIGID) {
        float* red12 = &s_part[0][0][0];          // (>= 16 * 12 floats; the posed-chunk scratch is free by now)
        hm_block_sum_n<12>(racc, red12);
        if (t < 12) hm_partial_store(out + 340 + t, racc[t]);
        __syncthreads();
    }
    if (t < nv) {
        const int v = v0 + t;
        float T[12];
        mano_skin_transform(m, sh, v, T);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s_g[t][c] = g[c];
            s_dvp[3 * t + c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
        }
        for (int j = 0; j < MANO_J; ++j) s_w[t][j] = m.weights[v * MANO_J + j];
    }
    MPH_MARK(1);
    const float tg0 = hm_block_sum(g[0], red), tg1 = hm_block_sum(g[1], red), tg2 = hm_block_sum(g[2], red);
    if (t == 0) { hm_partial_store(out + 337, tg0); hm_partial_store(out + 338, tg1); hm_partial_store(out + 339, tg2); }
    __syncthreads();
    MPH_MARK(2);
    // dA[j][r][c] = sum_v W[v][j] g[v][r] [vp;1][c]
    if (t < 192) {
        const int j = t / 12, r = (t % 12) / 4, c = t % 4;
        float acc = 0.f;
        for (int i = 0; i < nv; ++i) acc += s_w[i][j] * s_g[i][r] * (c < 3 ? s_vp[i][c] : 1.0f);
        hm_partial_store(out + t, acc);
    }
    __syncthreads();
    MPH_MARK(3);
    // dfeat[k] = sum_{v,c} M[k][3v+c] dvp[v][c]   (one wave per row, 3 coalesced loads, rows independent)
    const int wv = t >> 6, lane = t & 63;
    const int ne = 3 * nv;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (lane < ne) d0 = s_dvp[lane];
    if (lane + 64 < ne) d1 = s_dvp[lane + 64];
    if (lane + 128 < ne) d2 = s_dvp[lane + 128];
    // all the wave's rows are requested before the first one is reduced: the loop is otherwise one memory latency per
    // four rows with nothing else in flight (this loop was ~15 of the kernel's 36 us)
    constexpr int ROWS = (MANO_NF + 3) / 4;
    float r0[ROWS], r1[ROWS], r2[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int k = wv + 4 * q;
        const float* row = m.M + (long)min(k, MANO_NF - 1) * (3 * MANO_V) + 3 * v0;
        r0[q] = lane < ne ? row[lane] : 0.f;
        r1[q] = lane + 64 < ne ? row[lane + 64] : 0.f;
        r2[q] = lane + 128 < ne ? row[lane + 128] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int k = wv + 4 * q;
        float acc = r0[q] * d0;
        acc += r1[q] * d1;
        acc += r2[q] * d2;
        acc = hm_wave_sum(acc);
        if (lane == 0 && k < MANO_NF) hm_partial_store(out + 192 + k, acc);
    }
    // per-frame ticket: every thread's record stores must have landed before the workgroup takes it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MPH_MARK(4);
    if (t == 0) {
        const unsigned int ticket = atomicAdd(frame_cnt + lb, 1u);
        const int last = ticket == gridDim.x - 1u;
        if (last) atomicExch(frame_cnt + lb, 0u);
        s_flag = last;
    }
    __syncthreads();
    MPH_MARK(5);
    if (s_flag) {
        mano_bwd2_body(m, sh, w2, partials, gridDim.x, lb, b, pca_dim, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans);
        if (RIGID) {
            __syncthreads();
            if (t == 0) {              // the frame's dR (9) and dt (3), summed over the chunks in chunk order by the body above
                float dr6[6];
                rot6d_backward(mr.rot6d + (long)b * 6, &w2.tot[340], dr6);
#pragma unroll
                for (int k = 0; k < 6; ++k) mr.g_rot6d[(long)b * 6 + k] = dr6[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) mr.g_trans[(long)b * 3 + k] = w2.tot[349 + k];
            }
        }
#ifdef MANO_PHASES
        __syncthreads();
        MPH_MARK(6);
        if (t == 0) atomicAdd(&g_mano_ph[14], 1ull);
#endif
    }
}

extern "C" {
// model: 8 device pointers in the order of ManoModelDev.
// hm_mano_fwd_rows / hm_mano_bwd_rows: frame f of the launch is row row0 + f * row_stride of EVERY per-row array (parameters,
// gradients, vertices, state; arrays sized for B * row_stride rows) - the strided slice `i::hand_nb` of reference
// homan.py:343-358 without a copy.  clip_len counts rows.
int hm_mano_fwd_rows(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                     const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d,
                     const float* rigid_trans, const float* rigid_scale, float* verts_world, float* state, int clip_len,
                     int row0, int row_stride, hipStream_t stream)
{
    HM_CHECK_ARG(model && pca && rot && betas && verts && B > 0 && pca_dim >= 16 && row_stride >= 1 && row0 >= 0 && row0 < row_stride);
    HM_CHECK_ARG(clip_len >= 0 && (clip_len == 0 || (B * row_stride) % clip_len == 0));
    HM_CHECK_ARG(!verts_world || (rigid_rot6d && rigid_trans && rigid_scale));
    ManoModelDev m = {(const float*)model[0], (const float*)model[1], (const float*)model[2], (const float*)model[3],
                      (const float*)model[4], (const float*)model[5], (const float*)model[6], (const int*)model[7]};
    hipLaunchKernelGGL(k_mano_fwd, dim3(MANO_NCH64, (B + MANO_FPW - 1) / MANO_FPW), dim3(256), g_hm_lds_pad[HM_PAD_MANO_FWD], stream, m, pca, pca_dim, rot, betas, trans, B, verts,
                       joints, rigid_rot6d, rigid_trans, rigid_scale, verts_world, state, clip_len ? clip_len : B * row_stride,
                       row0, row_stride);
    return hm_launch_status();
}
int hm_mano_fwd_clips(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                      const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d,
                      const float* rigid_trans, const float* rigid_scale, float* verts_world, float* state, int clip_len,
                      hipStream_t stream)
{
    HM_CHECK_ARG(model && pca && rot && betas && verts && B > 0 && pca_dim >= 16 && HM_CLIP_LEN_OK(B, clip_len));
    HM_CHECK_ARG(!verts_world || (rigid_rot6d && rigid_trans && rigid_scale));
    ManoModelDev m = {(const float*)model[0], (const float*)model[1], (const float*)model[2], (const float*)model[3],
                      (const float*)model[4], (const float*)model[5], (const float*)model[6], (const int*)model[7]};
    hipLaunchKernelGGL(k_mano_fwd, dim3(MANO_NCH64, (B + MANO_FPW - 1) / MANO_FPW), dim3(256), g_hm_lds_pad[HM_PAD_MANO_FWD], stream, m, pca, pca_dim, rot, betas, trans, B, verts,
                       joints, rigid_rot6d, rigid_trans, rigid_scale, verts_world, state, clip_len ? clip_len : B, 0, 1);
    return hm_launch_status();
}
int hm_mano_fwd(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d, const float* rigid_trans,
                const float* rigid_scale, float* verts_world, float* state, hipStream_t stream)
{
    return hm_mano_fwd_clips(model, pca, pca_dim, rot, betas, trans, B, verts, joints, rigid_rot6d, rigid_trans,
                             rigid_scale, verts_world, state, 0, stream);
}
size_t hm_mano_workspace_bytes(int B) { return 512 + (size_t)B * 4 + (size_t)B * MANO_NCH64 * MANO_PART * sizeof(float); }
size_t hm_mano_state_bytes(int B) { return (size_t)B * MANO_STATE_DW * sizeof(float); }
// workspace: hm_mano_workspace_bytes(B), zero-filled once (per-frame tickets reset themselves).  state: the buffer the
// forward filled (hm_mano_state_bytes(B)) for the SAME parameters, or NULL to recompute the chain.
static int mano_bwd_launch(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                           const float* g_verts, const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot,
                           float* g_betas, float* g_trans, const float* state, void* workspace, int row0, int row_stride,
                           const ManoRigid& mr, hipStream_t stream)
{
    HM_CHECK_ARG(model && pca && rot && betas && (g_verts || mr.mesh) && g_pca && g_rot && g_betas && g_trans && workspace);
    HM_CHECK_ARG(B > 0 && pca_dim >= 16 && row_stride >= 1 && row0 >= 0 && row0 < row_stride);
    ManoModelDev m = {(const float*)model[0], (const float*)model[1], (const float*)model[2], (const float*)model[3],
                      (const float*)model[4], (const float*)model[5], (const float*)model[6], (const int*)model[7]};
    unsigned int* cnt = (unsigned int*)workspace;
    float* partials = (float*)((char*)workspace + 256 + (((size_t)B * 4 + 255) & ~(size_t)255));
    if (mr.mesh)
        hipLaunchKernelGGL(k_mano_bwd<true>, dim3(MANO_NCH64, B), dim3(256), g_hm_lds_pad[HM_PAD_MANO_BWD], stream, m, pca, pca_dim, rot, betas,
                           g_verts, B, state, partials, cnt, pca_dim, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans, row0,
                           row_stride, mr);
    else
        hipLaunchKernelGGL(k_mano_bwd<false>, dim3(MANO_NCH64, B), dim3(256), g_hm_lds_pad[HM_PAD_MANO_BWD], stream, m, pca, pca_dim, rot, betas,
                           g_verts, B, state, partials, cnt, pca_dim, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans, row0,
                           row_stride, mr);
    return hm_launch_status();
}
int hm_mano_bwd_rows(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                     const float* g_verts, const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                     float* g_trans, const float* state, void* workspace, int row0, int row_stride, hipStream_t stream)
{
    ManoRigid none;
    memset(&none, 0, sizeof(none));
    return mano_bwd_launch(model, pca, pca_dim, rot, betas, B, g_verts, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans,
                           state, workspace, row0, row_stride, none, stream);
}
// hm_rigid_bwd_clips of the hand (no mesh gradient buffer in between) + hm_mano_bwd in ONE launch: mesh = the forward's
// model-space vertices (B,778,3), rigid_* = the hand's rigid pose (scale: one per clip of clip_len frames, 0 = one clip),
// g_terms / weights / g_rigid / g_frame / frame_stride / frame_scale as hm_rigid_bwd_clips, g_rigid_rot6d (B,3,2) and
// g_rigid_trans (B,3) receive the rigid pose's gradients.  Same expressions as the two launches; the twelve sums of dR / dt
// are formed per vertex chunk and added in chunk order (last bits differ from hm_rigid_bwd_clips' 1024-thread sum).
int hm_mano_bwd_rigid_clips(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                            const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                            float* g_trans, const float* state, void* workspace, const float* mesh, const float* rigid_rot6d,
                            const float* rigid_scale, const float* const* g_terms, const float* weights, int n_terms,
                            const float* g_rigid, const float* g_frame, int frame_stride, float frame_scale,
                            float* g_rigid_rot6d, float* g_rigid_trans, int clip_len, hipStream_t stream)
{
    HM_CHECK_ARG(mesh && rigid_rot6d && rigid_scale && g_rigid_rot6d && g_rigid_trans && HM_CLIP_LEN_OK(B, clip_len));
    HM_CHECK_ARG(n_terms >= 0 && n_terms <= 5 && (n_terms == 0 || (g_terms && weights)));
    HM_CHECK_ARG(!g_frame || frame_stride >= 3);
    ManoRigid mr;
    memset(&mr, 0, sizeof(mr));
    mr.mesh = mesh; mr.rot6d = rigid_rot6d; mr.scale = rigid_scale; mr.clip_len = clip_len ? clip_len : B;
    for (int k = 0; k < 5; ++k) { mr.terms.p[k] = k < n_terms ? g_terms[k] : nullptr; mr.terms.w[k] = k < n_terms ? weights[k] : 0.f; }
    mr.g_rigid = g_rigid; mr.g_frame = g_frame; mr.frame_stride = frame_stride; mr.frame_scale = frame_scale;
    mr.g_rot6d = g_rigid_rot6d; mr.g_trans = g_rigid_trans;
    return mano_bwd_launch(model, pca, pca_dim, rot, betas, B, nullptr, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans,
                           state, workspace, 0, 1, mr, stream);
}
int hm_mano_bwd(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                const float* g_verts, const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                float* g_trans, const float* state, void* workspace, hipStream_t stream)
{
    return hm_mano_bwd_rows(model, pca, pca_dim, rot, betas, B, g_verts, g_pca_extra, w_extra, g_pca, g_rot, g_betas, g_trans,
                            state, workspace, 0, 1, stream);
}
#ifdef MANO_PHASES
int hm_debug_mano_phases(unsigned long long* out)
{
    unsigned long long z[16] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mano_ph), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mano_ph), z, sizeof(z));
    return HM_OK;
}
#endif
}  // extern "C"
