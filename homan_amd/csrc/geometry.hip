// geometry.hip -- batched 6-D rotation -> matrix and rigid vertex transform, forward + backward.
//
// Replaces reference homan/utils/geometry.py:9-27 (rot6d_to_matrix; cross product taken per row, see
// DESIGN.md for the reference's dim-less torch.cross quirk at batch==3) and
// homan/utils/camera.py:108-139 (compute_transformation_persp: (s*v) @ R + t and its mesh-detached twin)
// as used by homan/homan.py:298-307 (object) and :341-382 (hand).
#include "hm_common.h"

// verts[n,v,:] = (s * mesh[n,v,:]) @ R[n] + t[n]      grid (chunks, N)
__global__ __launch_bounds__(256) void k_rigid_fwd(const float* __restrict__ mesh, const float* __restrict__ rot6d,
                                                   const float* __restrict__ trans, const float* __restrict__ scale,
                                                   int abs_scale, int N, int V, float* __restrict__ rotmat,
                                                   float* __restrict__ verts, int clip_len)
{
    __shared__ float R[9];
    const int n = blockIdx.y;
    if (threadIdx.x == 0) {
        float r[9];
        rot6d_to_mat(rot6d + n * 6, r);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = r[k];
        if (blockIdx.x == 0 && rotmat)
#pragma unroll
            for (int k = 0; k < 9; ++k) rotmat[n * 9 + k] = r[k];
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float s = scale[n / clip_len];          // one scale per clip (clip = clip_len consecutive frames)
    if (abs_scale) s = fabsf(s);
    const float* m = mesh + ((long)n * V + v) * 3;
    const float x = s * m[0], y = s * m[1], z = s * m[2];
    float* o = verts + ((long)n * V + v) * 3;
    const float* t = trans + n * 3;
    o[0] = x * R[0] + y * R[3] + z * R[6] + t[0];
    o[1] = x * R[1] + y * R[4] + z * R[7] + t[1];
    o[2] = x * R[2] + y * R[5] + z * R[8] + t[2];
}

// backward.  g_full = sum_k w[k] * terms[k] (up to 4 weighted per-vertex gradients, NULL terms skipped) reaches mesh,
// scale, R, t ; g_rigid (per-vertex) and g_frame (one vector per frame, the same for every vertex) reach R, t only
// (gradients w.r.t. the mesh-detached twin of the vertices).  Summing the weighted terms here replaces a separate
// linear-combination launch on the critical chain.  grid (N)
// Optional fifth term: the silhouette gradient, gathered on the fly from the per-(face, corner) NDC gradients of the edge
// sweeps (hm_sil_bwd called with grad_verts == NULL) and pushed through the projection backward -- the work of
// k_bwd_gather, without its launch and without the (B,V,3) round trip on the critical chain.
struct SilGather {
    const double* parts;       // (B,F,3,2) exact per-corner sums (hm_sil_parts)
    const int* adj_off;        // (V+1) CSR over vertices
    const int* adj_items;      // face * 3 + corner
    const float* cam_verts;    // (B,V,3) camera-space vertices the silhouettes were rendered from
    const float* K;            // (B,3,3)
    float orig_size;
    int F;
};
// EXACT: the 13 per-frame sums over the vertices are order-independent (every addend rounded to the grid of `magic`, summed in
// double: hm_quant) - with the exact per-corner sums of the sweeps the object's pose gradients are then a function of the
// SET of terms, the same floats whatever the launch geometry, and the same as the CPU oracle's (oracle/csrc/objchain.c).
template <bool EXACT>
__global__ __launch_bounds__(1024) void k_rigid_bwd(const float* __restrict__ mesh, const float* __restrict__ rot6d,
                                                   const float* __restrict__ scale, int abs_scale, RigidTerms terms,
                                                   SilGather sil,
                                                   const float* __restrict__ g_rigid, const float* __restrict__ g_frame,
                                                   int frame_stride, float frame_scale, int N, int V,
                                                   float* __restrict__ g_mesh,
                                                   float* __restrict__ g_rot6d, float* __restrict__ g_trans,
                                                   float* __restrict__ g_scale_part, float* __restrict__ partials,
                                                   unsigned int* __restrict__ frame_cnt, int clip_len, double magic)
{
    HM_LATENCY_KERNEL();
    HM_STAMP_START(sil.parts ? 1 : 0);
    __shared__ float red13[16 * 13];
    __shared__ double red13d[EXACT ? 16 * 13 : 1];
    __shared__ int s_flag;
    const int n = blockIdx.x;
    // (every thread builds the frame's rotation itself from six uniform loads: no LDS hand-over, no barrier in front of the
    //  vertex loads - this kernel is a chain of dependent round trips on the tail of both streams)
    float R[9];
    rot6d_to_mat(rot6d + n * 6, R);
    const float sraw = scale[n / clip_len];
    const float s = abs_scale ? fabsf(sraw) : sraw;
    float gfr[3] = {0.f, 0.f, 0.f};
    if (g_frame) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gfr[c] = frame_scale * g_frame[(long)n * frame_stride + c];
    }
    float acc[13];
    double accd[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { acc[k] = 0.f; accd[k] = 0.0; }
    // grid (N, chunks): this workgroup's share of the vertices (up to 1024 threads x 4: ONE workgroup per frame for meshes of
    // <= 4096 vertices, no chunk records / ticket / second round trip); with several chunks the frame's last workgroup
    // (ticket) finishes the frame
    for (int v = blockIdx.y * blockDim.x + threadIdx.x; v < V; v += gridDim.y * blockDim.x) {
        const long o = ((long)n * V + v) * 3;
        const float m[3] = {mesh[o], mesh[o + 1], mesh[o + 2]};
        float gf[3] = {0.f, 0.f, 0.f}, gt[3];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (terms.p[k]) {
                gf[0] += terms.w[k] * terms.p[k][o]; gf[1] += terms.w[k] * terms.p[k][o + 1]; gf[2] += terms.w[k] * terms.p[k][o + 2];
            }
        if (sil.parts) {        // same arithmetic as k_bwd_gather (raster_sweep.hip)
            double su = 0.0, sv = 0.0;       // exact: the per-corner sums are multiples of one quantum
            const double2* pf = reinterpret_cast<const double2*>(sil.parts + (long)n * sil.F * 6);
            // eight adjacent corners at a time: all item loads, then all gradient loads (two dependent round trips per
            // batch instead of two per corner; the valence of a mesh vertex is ~6)
            const int a1 = sil.adj_off[v + 1];
            for (int a = sil.adj_off[v]; a < a1; a += 8) {
                int item[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) item[k] = a + k < a1 ? sil.adj_items[a + k] : -1;
                double2 g2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) g2[k] = item[k] >= 0 ? pf[item[k]] : make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < 8; ++k) { su += g2[k].x; sv += g2[k].y; }
            }
            const float gu = (float)su, gv = (float)sv;
            const float* k = sil.K + n * 9;
            const float x = sil.cam_verts[o], y = sil.cam_verts[o + 1], z = sil.cam_verts[o + 2];
            const float zz = z + 1e-9f;
            const float du0 = gu * (2.0f / sil.orig_size), dv0 = -gv * (2.0f / sil.orig_size);
            const float dxn = k[0] * du0 + k[3] * dv0;
            const float dyn = k[1] * du0 + k[4] * dv0;
            gf[0] += dxn / zz;
            gf[1] += dyn / zz;
            gf[2] += -(dxn * x + dyn * y) / (zz * zz);
        }
        gt[0] = gf[0] + gfr[0]; gt[1] = gf[1] + gfr[1]; gt[2] = gf[2] + gfr[2];
        if (g_rigid) { gt[0] += g_rigid[o]; gt[1] += g_rigid[o + 1]; gt[2] += g_rigid[o + 2]; }
        // d(s*m)_i = sum_j R[i][j] gf_j
        const float dm[3] = {R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2], R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2],
                             R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]};
        if (EXACT) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) accd[3 * i + j] += hm_quant((s * m[i]) * gt[j], magic);
#pragma unroll
            for (int j = 0; j < 3; ++j) accd[9 + j] += hm_quant(gt[j], magic);
            accd[12] += hm_quant(m[0] * dm[0] + m[1] * dm[1] + m[2] * dm[2], magic);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[3 * i + j] += (s * m[i]) * gt[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[9 + j] += gt[j];
            acc[12] += m[0] * dm[0] + m[1] * dm[1] + m[2] * dm[2];
        }
        if (g_mesh) { g_mesh[o] = s * dm[0]; g_mesh[o + 1] = s * dm[1]; g_mesh[o + 2] = s * dm[2]; }
    }
    float tot[13];
    if (EXACT) {
        hm_block_sum_n_f64<13>(accd, red13d);
        if (gridDim.y > 1) {
            // chunk records of 13 doubles (32 floats apart), agent-scope stores / loads like the float records; the frame's last
            // workgroup adds them - in any order: the sums are exact
            double* rec = reinterpret_cast<double*>(partials + ((long)n * gridDim.y + blockIdx.y) * 32);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < 13; ++k) __hip_atomic_store(rec + k, accd[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!hm_last_block(frame_cnt + n, gridDim.y, &s_flag)) return;
            if (threadIdx.x < 13 * gridDim.y)
                red13d[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const double*>(partials + ((long)n * gridDim.y + threadIdx.x / 13) * 32) + threadIdx.x % 13,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < 13; ++k) accd[k] = 0.0;
                for (unsigned c = 0; c < gridDim.y; ++c)
#pragma unroll
                    for (int k = 0; k < 13; ++k) accd[k] += red13d[c * 13 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) tot[k] = (float)accd[k];
    } else {
#pragma unroll
    for (int k = 0; k < 13; ++k) tot[k] = acc[k];
    hm_block_sum_n<13>(tot, red13);
    if (gridDim.y > 1) {
        float* rec = partials + ((long)n * gridDim.y + blockIdx.y) * 32;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) hm_partial_store(rec + k, tot[k]);
        }
        if (!hm_last_block(frame_cnt + n, gridDim.y, &s_flag)) return;
        // all records in ONE round trip (a thread per value), then the sums in chunk order: deterministic, and the frame's
        // tail is one memory latency instead of one per chunk (<= 16 chunks x 13 values fit red13)
        if (threadIdx.x < 13 * gridDim.y)
            red13[threadIdx.x] = hm_partial_load(partials + ((long)n * gridDim.y + threadIdx.x / 13) * 32 + threadIdx.x % 13);
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) tot[k] = 0.f;
            for (unsigned c = 0; c < gridDim.y; ++c)       // fixed order
#pragma unroll
                for (int k = 0; k < 13; ++k) tot[k] += red13[c * 13 + k];
        }
    }
    }
    if (threadIdx.x == 0) {
        float dr6[6];
        rot6d_backward(rot6d + n * 6, tot, dr6);
#pragma unroll
        for (int k = 0; k < 6; ++k) g_rot6d[n * 6 + k] = dr6[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_trans[n * 3 + k] = tot[9 + k];
        if (g_scale_part) g_scale_part[n] = (abs_scale && sraw < 0.f) ? -tot[12] : tot[12];
    }
    HM_STAMP_END(sil.parts ? 1 : 0);
}

// The object's rigid backward of the fused loops (exact sums, silhouette gather): one workgroup of up to 1024 threads per
// (frame, chunk of 1024 * NV vertices), every thread NV vertices whose dependent loads - CSR offsets -> corner items ->
// per-corner sums - are issued STAGE BY STAGE for all NV at once: the kernel is a chain of round trips on the tail of the
// iteration, and a loop over a thread's vertices walked that chain once per vertex.  Meshes of <= 1024 * NV vertices (the
// 1502-vertex bottle at NV = 2) are one workgroup per frame: no chunk records, no ticket, no second round trip.
// Optional temporal-smoothness term of the SAME vertices (reference homan/lossutils.py:18-36), formed here from the
// camera-space vertices of the neighbouring frames with the arithmetic of smooth_body (pair_bodies.h): the object's chain then
// waits for nothing the hand-side stream produces.  It is added FIRST, like the first entry of `terms` used to be.
struct SmoothIn { const float* verts; float w; };
template <int NV, int MAXT>
__global__ __launch_bounds__(MAXT) void k_rigid_bwd_x(const float* __restrict__ mesh, const float* __restrict__ rot6d,
                                                     const float* __restrict__ scale, int abs_scale, RigidTerms terms,
                                                     SilGather sil, SmoothIn sm, int N, int V, float* __restrict__ g_rot6d,
                                                     float* __restrict__ g_trans, float* __restrict__ g_scale_part,
                                                     float* __restrict__ partials, unsigned int* __restrict__ frame_cnt,
                                                     int clip_len, double magic)
{
    HM_LATENCY_KERNEL();
    HM_STAMP_START(1);
    __shared__ double red13d[16 * 13];
    __shared__ int s_flag;
    const int n = blockIdx.x;
    float R[9];
    rot6d_to_mat(rot6d + n * 6, R);
    const float sraw = scale[n / clip_len];
    const float s = abs_scale ? fabsf(sraw) : sraw;
    const int fl = n % clip_len;                                    // frame inside its clip (smoothness: no term across clips)
    const long row = (long)V * 3;
    const long scnt = (long)(clip_len - 1) * row;
    const float inv_cnt = scnt > 0 ? 1.0f / (float)scnt : 0.f;
    int vv[NV];
    bool ok[NV];
    // ---- stage A: everything addressed by the vertex alone
    int a0[NV], a1[NV];
    float m[NV][3], cv[NV][3], tv[NV][5][3], nb[NV][2][3];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        vv[j] = ((int)blockIdx.y * NV + j) * (int)blockDim.x + (int)threadIdx.x;
        ok[j] = vv[j] < V;
        const int v = ok[j] ? vv[j] : 0;
        const long o = ((long)n * V + v) * 3;
        a0[j] = sil.parts ? sil.adj_off[v] : 0;
        a1[j] = sil.parts ? sil.adj_off[v + 1] : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m[j][c] = mesh[o + c];
            cv[j][c] = sil.parts ? sil.cam_verts[o + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) tv[j][k][c] = terms.p[k] ? terms.p[k][o + c] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            nb[j][0][c] = (sm.verts && fl + 1 < clip_len) ? sm.verts[o + row + c] : 0.f;
            nb[j][1][c] = (sm.verts && fl >= 1) ? sm.verts[o - row + c] : 0.f;
        }
    }
    float sv[NV][3];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) sv[j][c] = sm.verts ? sm.verts[((long)n * V + (ok[j] ? vv[j] : 0)) * 3 + c] : 0.f;
    // ---- stage B: the first eight corner items of every vertex; stage C: their per-corner sums
    int item[NV][8];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) item[j][k] = (ok[j] && a0[j] + k < a1[j]) ? sil.adj_items[a0[j] + k] : -1;
    double su[NV], sw[NV];
    {
        const double2* pf = reinterpret_cast<const double2*>(sil.parts + (long)n * sil.F * 6);
        double2 g2[NV][8];
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) g2[j][k] = item[j][k] >= 0 ? pf[item[j][k]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            su[j] = 0.0; sw[j] = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { su[j] += g2[j][k].x; sw[j] += g2[j][k].y; }
            for (int a = a0[j] + 8; ok[j] && a < a1[j]; ++a) {          // (valence > 8: rare)
                const double2 g = pf[sil.adj_items[a]];
                su[j] += g.x; sw[j] += g.y;
            }
        }
    }
    double accd[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) accd[k] = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (!ok[j]) continue;
        float gf[3] = {0.f, 0.f, 0.f}, gt[3];
        if (sm.verts) {         // smooth_body: g = -(v[t+1] - v) + (v - v[t-1]); unit = 2 g / count
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float g = 0.f;
                if (fl + 1 < clip_len) { const float d = nb[j][0][c] - sv[j][c]; g -= d; }
                if (fl >= 1) g += sv[j][c] - nb[j][1][c];
                gf[c] += sm.w * (2.0f * g * inv_cnt);
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (terms.p[k]) {
                gf[0] += terms.w[k] * tv[j][k][0]; gf[1] += terms.w[k] * tv[j][k][1]; gf[2] += terms.w[k] * tv[j][k][2];
            }
        if (sil.parts) {        // the arithmetic of k_bwd_gather (raster_sweep.hip)
            const float gu = (float)su[j], gv = (float)sw[j];
            const float* k = sil.K + n * 9;
            const float x = cv[j][0], y = cv[j][1], z = cv[j][2];
            const float zz = z + 1e-9f;
            const float du0 = gu * (2.0f / sil.orig_size), dv0 = -gv * (2.0f / sil.orig_size);
            const float dxn = k[0] * du0 + k[3] * dv0;
            const float dyn = k[1] * du0 + k[4] * dv0;
            gf[0] += dxn / zz;
            gf[1] += dyn / zz;
            gf[2] += -(dxn * x + dyn * y) / (zz * zz);
        }
        gt[0] = gf[0] + 0.f; gt[1] = gf[1] + 0.f; gt[2] = gf[2] + 0.f;
        const float dm[3] = {R[0] * gf[0] + R[1] * gf[1] + R[2] * gf[2], R[3] * gf[0] + R[4] * gf[1] + R[5] * gf[2],
                             R[6] * gf[0] + R[7] * gf[1] + R[8] * gf[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) accd[3 * i + q] += hm_quant((s * m[j][i]) * gt[q], magic);
#pragma unroll
        for (int q = 0; q < 3; ++q) accd[9 + q] += hm_quant(gt[q], magic);
        accd[12] += hm_quant(m[j][0] * dm[0] + m[j][1] * dm[1] + m[j][2] * dm[2], magic);
    }
    hm_block_sum_n_f64<13>(accd, red13d);
    if (gridDim.y > 1) {
        double* rec = reinterpret_cast<double*>(partials + ((long)n * gridDim.y + blockIdx.y) * 32);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) __hip_atomic_store(rec + k, accd[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!hm_last_block(frame_cnt + n, gridDim.y, &s_flag)) return;
        if (threadIdx.x < 13 * gridDim.y)
            red13d[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const double*>(partials + ((long)n * gridDim.y + threadIdx.x / 13) * 32) + threadIdx.x % 13,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) accd[k] = 0.0;
            for (unsigned c = 0; c < gridDim.y; ++c)
#pragma unroll
                for (int k = 0; k < 13; ++k) accd[k] += red13d[c * 13 + k];
        }
    }
    if (threadIdx.x == 0) {
        float tot[13], dr6[6];
#pragma unroll
        for (int k = 0; k < 13; ++k) tot[k] = (float)accd[k];
        rot6d_backward(rot6d + n * 6, tot, dr6);
#pragma unroll
        for (int k = 0; k < 6; ++k) g_rot6d[n * 6 + k] = dr6[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_trans[n * 3 + k] = tot[9 + k];
        if (g_scale_part) g_scale_part[n] = (abs_scale && sraw < 0.f) ? -tot[12] : tot[12];
    }
    HM_STAMP_END(1);
}

// out[i] = s[0] * in[i]   (backward of "loss = f(x)" ops whose unit gradient was produced in the forward)
__global__ void k_scale_by(const float* __restrict__ in, const float* __restrict__ s, long n, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = s[0] * in[i];
}
// out[i] = s0[0]*a[i] + s1[0]*b[i]
__global__ void k_scale2_by(const float* __restrict__ a, const float* __restrict__ s0, const float* __restrict__ b,
                            const float* __restrict__ s1, long n, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = s0[0] * a[i] + s1[0] * b[i];
}

// out[i] = w0*a0[i] + w1*a1[i] + w2*a2[i] + w3*a3[i]   (NULL terms are skipped; weights are host constants)
__global__ void k_lincomb4(const float* __restrict__ a0, float w0, const float* __restrict__ a1, float w1,
                           const float* __restrict__ a2, float w2, const float* __restrict__ a3, float w3, long n,
                           float* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    if (a0) v += w0 * a0[i];
    if (a1) v += w1 * a1[i];
    if (a2) v += w2 * a2[i];
    if (a3) v += w3 * a3[i];
    out[i] = v;
}
// out[c] = w0 * sum(parts[c*n .. c*n+n)) + w1 * extra[c]      (scale gradients: per-frame partials + prior term)
// grid (clips)
__global__ void k_sum_small(const float* __restrict__ parts, int n, float w0, const float* __restrict__ extra, float w1,
                            float* __restrict__ out)
{
    __shared__ float red[16];
    const int c = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += parts[(long)c * n + i];
    a = hm_block_sum(a, red);
    if (threadIdx.x == 0) out[c] = w0 * a + (extra ? w1 * extra[c] : 0.f);
}

thread_local int g_hm_lds_pad[HM_PAD_FAMILIES] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" {
int hm_lincomb4(const float* a0, float w0, const float* a1, float w1, const float* a2, float w2, const float* a3,
                float w3, long n, float* out, hipStream_t stream)
{
    HM_CHECK_ARG(out && n > 0);
    hipLaunchKernelGGL(k_lincomb4, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, a0, w0, a1, w1, a2, w2, a3, w3, n, out);
    return hm_launch_status();
}
int hm_sum_small_clips(const float* parts, int n, float w0, const float* extra, float w1, float* out, int nclips,
                       hipStream_t stream)
{
    HM_CHECK_ARG(parts && out && n > 0 && nclips > 0);
    hipLaunchKernelGGL(k_sum_small, dim3(nclips), dim3(64), 0, stream, parts, n, w0, extra, w1, out);
    return hm_launch_status();
}
int hm_sum_small(const float* parts, int n, float w0, const float* extra, float w1, float* out, hipStream_t stream)
{
    return hm_sum_small_clips(parts, n, w0, extra, w1, out, 1, stream);
}
int hm_rigid_fwd_clips(const float* mesh, const float* rot6d, const float* trans, const float* scale, int abs_scale, int N,
                       int V, float* rotmat, float* verts, int clip_len, hipStream_t stream)
{
    HM_CHECK_ARG(mesh && rot6d && trans && scale && verts && N > 0 && V > 0 && HM_CLIP_LEN_OK(N, clip_len));
    hipLaunchKernelGGL(k_rigid_fwd, dim3(hm_cdiv(V, 256), N), dim3(256), 0, stream, mesh, rot6d, trans, scale,
                       abs_scale, N, V, rotmat, verts, clip_len ? clip_len : N);
    return hm_launch_status();
}
int hm_rigid_fwd(const float* mesh, const float* rot6d, const float* trans, const float* scale, int abs_scale, int N,
                 int V, float* rotmat, float* verts, hipStream_t stream)
{
    return hm_rigid_fwd_clips(mesh, rot6d, trans, scale, abs_scale, N, V, rotmat, verts, 0, stream);
}
// see hm_common.h: family 0 MANO forward, 1 MANO backward, 2 smoothness / interaction / hand-terms launches, 3 the fused
// pair-terms launch, 4 rigid backward.  Per calling thread (thread-local), read at launch (or capture).  Returns the previous value; bytes < 0 queries.
int hm_tune_lds_pad(int family, int bytes)
{
    if (family < 0 || family >= HM_PAD_FAMILIES) return -1;
    const int prev = g_hm_lds_pad[family];
    if (bytes >= 0) g_hm_lds_pad[family] = bytes;
    return prev;
}
#define RIGID_MAX_CHUNKS 16
// Scheduling hint, no effect on results (the sums are exact): 1 = the object's rigid backward of the fused loops as ceil(V / 256)
// small workgroups per frame + ticket instead of one large workgroup per frame.  Per calling thread (thread-local), read at launch (or capture).
// Returns the previous value; < 0 only queries.
static thread_local int g_rigid_chunked = 1;      // (same-box A/B, cfg2: chunked 6295 it/s, one 768-thread workgroup per frame 6110)
int hm_tune_rigid_chunked(int enable)
{
    const int prev = g_rigid_chunked;
    if (enable >= 0) g_rigid_chunked = enable ? 1 : 0;
    return prev;
}
size_t hm_rigid_workspace_bytes(int N) { return (((size_t)N * 4 + 255) & ~(size_t)255) + (size_t)N * RIGID_MAX_CHUNKS * 32 * 4; }
static int rigid_bwd_launch(const float* mesh, const float* rot6d, const float* scale, int abs_scale,
                            const float* const* g_terms, const float* weights, int n_terms, SilGather sil,
                            const float* g_rigid, const float* g_frame, int frame_stride, float frame_scale, int N, int V,
                            float* g_mesh, float* g_rot6d, float* g_trans, float* g_scale_part, void* workspace,
                            int clip_len, int exact, int sum_log2q, SmoothIn smooth, hipStream_t stream)
{
    HM_CHECK_ARG(sum_log2q <= 0 && sum_log2q >= -60);
    HM_CHECK_ARG(mesh && rot6d && scale && g_rot6d && g_trans && N > 0 && V > 0 && HM_CLIP_LEN_OK(N, clip_len));
    HM_CHECK_ARG(n_terms >= 0 && n_terms <= 5 && (n_terms == 0 || (g_terms && weights)));
    HM_CHECK_ARG(!g_frame || frame_stride >= 3);
    RigidTerms t;
    for (int k = 0; k < 5; ++k) { t.p[k] = k < n_terms ? g_terms[k] : nullptr; t.w[k] = k < n_terms ? weights[k] : 0.f; }
    // workspace (hm_rigid_workspace_bytes, zero-filled once): per-frame tickets + chunk partials -> grid (N, chunks);
    // without it one workgroup per frame does everything
    // exact sums: any split of the vertices gives the same result, so the frame goes to ceil(V / 256) small workgroups (one
    // vertex per thread: one pass through the chain of dependent loads instead of one per vertex of a thread's share)
    const int threads = exact ? 256 : (V > 512 ? 1024 : 256);
    const int chunks = workspace ? min(RIGID_MAX_CHUNKS, hm_cdiv(V, exact ? threads : 4 * threads)) : 1;
    unsigned int* cnt = (unsigned int*)workspace;
    float* partials = workspace ? (float*)((char*)workspace + (((size_t)N * 4 + 255) & ~(size_t)255)) : nullptr;
    // one workgroup per frame whenever the mesh fits: <= 1024 vertices one per thread, <= 1536 two per thread on 768 threads (12
    // waves: 170 registers each, the 64 of the staged double2 loads included - at 1024 threads the 128-register cap spilled)
    // (meshes beyond 16 chunks of 256 vertices: 768 threads x 2 vertices per chunk, up to 24 576 vertices)
    const bool small = g_rigid_chunked && V <= RIGID_MAX_CHUNKS * 256;
    const int nv = small ? 1 : (V > 1024 ? 2 : 1);
    const int thr = small ? 256 : (nv == 2 ? 768 : (V > 512 ? 1024 : (V > 256 ? 512 : 256)));
    if (exact && !g_rigid && !g_frame && !g_mesh && (workspace ? V <= RIGID_MAX_CHUNKS * thr * nv : V <= thr * nv)) {
        // (the fused loops' object chain: stage-wise loads of NV vertices per thread, see k_rigid_bwd_x)
        const int ch = hm_cdiv(V, thr * nv);
        const double magic = hm_sum_magic(sum_log2q);
        if (nv == 2)
            hipLaunchKernelGGL((k_rigid_bwd_x<2, 768>), dim3(N, ch), dim3(thr), g_hm_lds_pad[HM_PAD_RIGID_BWD], stream, mesh, rot6d, scale,
                               abs_scale, t, sil, smooth, N, V, g_rot6d, g_trans, g_scale_part, partials, cnt, clip_len ? clip_len : N, magic);
        else
            hipLaunchKernelGGL((k_rigid_bwd_x<1, 1024>), dim3(N, ch), dim3(thr), g_hm_lds_pad[HM_PAD_RIGID_BWD], stream, mesh, rot6d, scale,
                               abs_scale, t, sil, smooth, N, V, g_rot6d, g_trans, g_scale_part, partials, cnt, clip_len ? clip_len : N, magic);
        return hm_launch_status();
    }
    HM_CHECK_ARG(!smooth.verts);
    if (exact)
        hipLaunchKernelGGL(k_rigid_bwd<true>, dim3(N, chunks), dim3(threads), g_hm_lds_pad[HM_PAD_RIGID_BWD], stream, mesh, rot6d, scale,
                           abs_scale, t, sil, g_rigid, g_frame, frame_stride, frame_scale, N, V, g_mesh, g_rot6d, g_trans,
                           g_scale_part, partials, cnt, clip_len ? clip_len : N, hm_sum_magic(sum_log2q));
    else
        hipLaunchKernelGGL(k_rigid_bwd<false>, dim3(N, chunks), dim3(threads), g_hm_lds_pad[HM_PAD_RIGID_BWD], stream, mesh, rot6d, scale,
                           abs_scale, t, sil, g_rigid, g_frame, frame_stride, frame_scale, N, V, g_mesh, g_rot6d, g_trans,
                           g_scale_part, partials, cnt, clip_len ? clip_len : N, 0.0);
    return hm_launch_status();
}
int hm_rigid_bwd_clips(const float* mesh, const float* rot6d, const float* scale, int abs_scale,
                       const float* const* g_terms, const float* weights, int n_terms, const float* g_rigid,
                       const float* g_frame, int frame_stride, float frame_scale, int N, int V, float* g_mesh,
                       float* g_rot6d, float* g_trans, float* g_scale_part, void* workspace, int clip_len,
                       hipStream_t stream)
{
    SilGather none = {nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, 0};
    return rigid_bwd_launch(mesh, rot6d, scale, abs_scale, g_terms, weights, n_terms, none, g_rigid, g_frame, frame_stride,
                            frame_scale, N, V, g_mesh, g_rot6d, g_trans, g_scale_part, workspace, clip_len, 0, 0,
                            SmoothIn{nullptr, 0.f}, stream);
}
int hm_rigid_bwd(const float* mesh, const float* rot6d, const float* scale, int abs_scale, const float* const* g_terms,
                 const float* weights, int n_terms, const float* g_rigid, const float* g_frame, int frame_stride,
                 float frame_scale, int N, int V, float* g_mesh, float* g_rot6d, float* g_trans, float* g_scale_part,
                 void* workspace, hipStream_t stream)
{
    return hm_rigid_bwd_clips(mesh, rot6d, scale, abs_scale, g_terms, weights, n_terms, g_rigid, g_frame, frame_stride,
                              frame_scale, N, V, g_mesh, g_rot6d, g_trans, g_scale_part, workspace, 0, stream);
}
// hm_rigid_bwd with the silhouette gradient as an extra full term taken straight from the sweep output: sil_parts =
// hm_sil_parts(workspace) of an hm_sil_bwd called with grad_verts == NULL; cam_verts / K / orig_size / F as given to it.
// The per-frame sums over the vertices are exact sums on the grid 2^sum_log2q (0: the default, 2^-44) like the per-corner
// sums they start from: the pose gradients do not depend on the launch geometry.
// smooth_verts (optional): the camera-space vertices (N,V,3) of the same mesh; the gradient of smooth_weight *
// mean((v[t+1] - v[t])^2) over every clip's frames (reference homan/lossutils.py:18-36) is formed inside the launch and added
// BEFORE `g_terms` - the values hm_smooth_fwd_clips' unit gradient times the weight would give as first term.
int hm_rigid_bwd_sil_clips(const float* mesh, const float* rot6d, const float* scale, int abs_scale,
                           const float* const* g_terms, const float* weights, int n_terms, const double* sil_parts,
                           const int* adj_off, const int* adj_items, const float* cam_verts, const float* K,
                           float orig_size, int F, int N, int V, float* g_rot6d, float* g_trans, float* g_scale_part,
                           void* workspace, int clip_len, int sum_log2q, const float* smooth_verts, float smooth_weight,
                           hipStream_t stream)
{
    HM_CHECK_ARG(sil_parts && adj_off && adj_items && cam_verts && K && F > 0);
    SilGather sil = {sil_parts, adj_off, adj_items, cam_verts, K, orig_size, F};
    return rigid_bwd_launch(mesh, rot6d, scale, abs_scale, g_terms, weights, n_terms, sil, nullptr, nullptr, 0, 0.f, N, V,
                            nullptr, g_rot6d, g_trans, g_scale_part, workspace, clip_len, 1, sum_log2q,
                            SmoothIn{smooth_verts, smooth_weight}, stream);
}
int hm_rigid_bwd_sil(const float* mesh, const float* rot6d, const float* scale, int abs_scale, const float* const* g_terms,
                     const float* weights, int n_terms, const double* sil_parts, const int* adj_off, const int* adj_items,
                     const float* cam_verts, const float* K, float orig_size, int F, int N, int V, float* g_rot6d,
                     float* g_trans, float* g_scale_part, void* workspace, int sum_log2q, hipStream_t stream)
{
    return hm_rigid_bwd_sil_clips(mesh, rot6d, scale, abs_scale, g_terms, weights, n_terms, sil_parts, adj_off, adj_items,
                                  cam_verts, K, orig_size, F, N, V, g_rot6d, g_trans, g_scale_part, workspace, 0, sum_log2q,
                                  nullptr, 0.f, stream);
}
int hm_scale_by(const float* in, const float* s, long n, float* out, hipStream_t stream)
{
    HM_CHECK_ARG(in && s && out && n > 0);
    hipLaunchKernelGGL(k_scale_by, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, in, s, n, out);
    return hm_launch_status();
}
int hm_scale2_by(const float* a, const float* s0, const float* b, const float* s1, long n, float* out,
                 hipStream_t stream)
{
    HM_CHECK_ARG(a && b && s0 && s1 && out && n > 0);
    hipLaunchKernelGGL(k_scale2_by, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, a, s0, b, s1, n, out);
    return hm_launch_status();
}
#ifdef HM_CHAIN_STAMPS
int hm_debug_chain_geometry(unsigned long long* out, int reset)
{
    unsigned long long z[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    (void)hipDeviceSynchronize();
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), z, sizeof(z));
    return HM_OK;
}
#endif
}  // extern "C"
