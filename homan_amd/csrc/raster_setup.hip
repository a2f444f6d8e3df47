// raster_setup.hip -- face setup of the silhouette rasteriser (projection, sample boxes, super-region bins)
#include "raster_ws.h"

// ---------------------------------------------------------------- projection (nr.projection, zero distortion)
// camera-space vertex, K (3,3) -> (u, v, z), u,v in [-1,1], v up.
__device__ __forceinline__ void project_vertex(const float* __restrict__ p, const float* __restrict__ k, float orig_size,
                                               float* out)
{
    const float x = p[0], y = p[1], z = p[2];
    const float zz = z + 1e-9f;
    const float xn = x / zz, yn = y / zz;
    float u = xn * k[0] + yn * k[1];
    u = u + k[2];
    float v = xn * k[3] + yn * k[4];
    v = v + k[5];
    v = orig_size - v;
    u = 2.0f * (u - orig_size / 2.0f) / orig_size;
    v = 2.0f * (v - orig_size / 2.0f) / orig_size;
    out[0] = u; out[1] = v; out[2] = z;
}

// ---------------------------------------------------------------- face setup
// projects the three vertices of every face (a vertex is shared by ~6 faces: re-projecting it is cheaper than a
// separate projection launch on the critical path), packs the (B,F,3,3) NDC face buffer and the 8-byte screen boxes.
// It also bins the faces into super-regions of 64x64 or 128x128 samples (hm_sr_shift) (counts aggregated per workgroup in LDS, one global atomic per
// (workgroup, bin)): a raster workgroup then scans the faces of its super-region instead of the whole frame.  The order
// inside a bin is arbitrary; the raster resolves visibility with a min, so its result does not depend on it.
// grid (ceil(F/256), B).  bin_cnt must be zero on entry (the raster's last workgroup resets it).
// (body: bx / by = the workgroup's block index within its render / its frame; every argument is workgroup-uniform)
__device__ __forceinline__ void setup_faces_body(const int bx, const int by, const float* __restrict__ verts, const float* __restrict__ K,
                                                     float orig_size, const int* __restrict__ faces, int faces_bstride,
                                                     int B, int V, int F, int is, float* __restrict__ faces9,
                                                     FaceBox* __restrict__ boxes, unsigned char* __restrict__ owned,
                                                     int* __restrict__ bin_cnt, int* __restrict__ bin_list,
                                                     const float* __restrict__ rigid_rot6d,
                                                     const float* __restrict__ rigid_trans,
                                                     const float* __restrict__ rigid_scale, int rigid_abs, int clip_len,
                                                     float* __restrict__ cam_out, int nfb)
{
    HM_STAMP_START(0);
    __shared__ int s_cnt[SR_MAX], s_base[SR_MAX];
    __shared__ float s_R[9];
    // optional rigid transform of mesh-space `verts` (same arithmetic as hm_rigid_fwd, so the camera-space vertices the
    // other losses get from that entry point are the very numbers rasterised here): the silhouette chain then does not
    // wait for a separate transform launch
    if (rigid_rot6d && threadIdx.x == 0) rot6d_to_mat(rigid_rot6d + by * 6, s_R);
    if (bx >= nfb) {
        // vertex blocks behind the face blocks (cam_out != NULL): the camera-space vertices themselves, for the caller's
        // other losses - the arithmetic of k_rigid_fwd, so hm_rigid_fwd on the same inputs returns the same floats, and the
        // caller's second stream no longer opens with a transform launch of its own
        __syncthreads();
        const int bb = by, v = (bx - nfb) * blockDim.x + threadIdx.x;
        if (v >= V) return;
        float sc = rigid_scale[bb / clip_len];
        if (rigid_abs) sc = fabsf(sc);
        const float* m = verts + ((long)bb * V + v) * 3;
        const float x = sc * m[0], y = sc * m[1], z = sc * m[2];
        const float* t = rigid_trans + bb * 3;
        float* o = cam_out + ((long)bb * V + v) * 3;
        o[0] = x * s_R[0] + y * s_R[3] + z * s_R[6] + t[0];
        o[1] = x * s_R[1] + y * s_R[4] + z * s_R[7] + t[1];
        o[2] = x * s_R[2] + y * s_R[5] + z * s_R[8] + t[2];
        return;
    }
    const int b = by, fi = bx * blockDim.x + threadIdx.x;
    const bool valid = fi < F;
    const int nsx = (is + (1 << hm_sr_shift(is)) - 1) >> hm_sr_shift(is), nsr = nsx * nsx;
    if (threadIdx.x < SR_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned mask = 0;
    int x0 = 1, y0 = 1, x1 = 0, y1 = 0;
    if (valid) {
        const long i = (long)b * F + fi;
        const int* fc = faces + (long)b * faces_bstride + 3 * fi;
        float f[9], r[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* mv = verts + ((long)b * V + fc[k]) * 3;
            float cam[3] = {mv[0], mv[1], mv[2]};
            if (rigid_rot6d) {
                float sc = rigid_scale[b / clip_len];      // one object scale per clip
                if (rigid_abs) sc = fabsf(sc);
                const float x = sc * mv[0], y = sc * mv[1], z = sc * mv[2];
                const float* t = rigid_trans + b * 3;
                cam[0] = x * s_R[0] + y * s_R[3] + z * s_R[6] + t[0];
                cam[1] = x * s_R[1] + y * s_R[4] + z * s_R[7] + t[1];
                cam[2] = x * s_R[2] + y * s_R[5] + z * s_R[8] + t[2];
            }
            project_vertex(cam, K + b * 9, orig_size, f + 3 * k);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { r[3 * k] = f[3 * (2 - k)]; r[3 * k + 1] = f[3 * (2 - k) + 1]; r[3 * k + 2] = f[3 * (2 - k) + 2]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) faces9[i * 9 + k] = f[k];
        owned[(long)b * 2 * F + fi] = 0;          // "owns at least one sample" flags, set by the forward raster
        owned[(long)b * 2 * F + F + fi] = 0;
        mask = (backside(f) ? 0u : 1u) | (backside(r) ? 0u : 2u);
        float px[3], py[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { px[k] = topix(f[3 * k], is); py[k] = topix(f[3 * k + 1], is); }
        const float xmin = fminf(px[0], fminf(px[1], px[2])), xmax = fmaxf(px[0], fmaxf(px[1], px[2]));
        const float ymin = fminf(py[0], fminf(py[1], py[2])), ymax = fmaxf(py[0], fmaxf(py[1], py[2]));
        if (!(xmax >= -2.0f && ymax >= -2.0f && xmin <= is + 1.0f && ymin <= is + 1.0f)) mask = 0;  // off-screen / NaN
        // a vertex projected beyond 1e15 (a camera-space depth within 1e-15 of the image plane), or not finite: the edge
        // functions of such a face overflow to inf - inf; culled here and in the oracle (oracle/csrc/nmr_raster.c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(fabsf(f[3 * k]) <= 1e15f && fabsf(f[3 * k + 1]) <= 1e15f)) mask = 0;
        // sample p (integer pixel coordinate) can be covered only if min <= p <= max; 0.01 px of slack dwarfs the
        // rounding of the edge functions, so the box is conservative yet tight
        x0 = max(0, (int)ceilf(fmaxf(xmin, -2.0f) - 0.01f));
        x1 = min(is - 1, (int)floorf(fminf(xmax, is + 1.0f) + 0.01f));
        y0 = max(0, (int)ceilf(fmaxf(ymin, -2.0f) - 0.01f));
        y1 = min(is - 1, (int)floorf(fminf(ymax, is + 1.0f) + 0.01f));
        if (x1 < x0 || y1 < y0) mask = 0;
        if (mask == 0) { x0 = y0 = 1; x1 = y1 = 0; }
        FaceBox bx;
        bx.x0m = (unsigned short)(x0 | (mask << 14));
        bx.y0 = (unsigned short)y0;
        bx.x1 = (unsigned short)x1;
        bx.y1 = (unsigned short)y1;
        boxes[i] = bx;
    }
    if (!bin_cnt) return;
    // local slots in LDS, one global reservation per (workgroup, bin)
    const int sx0 = x0 >> hm_sr_shift(is), sx1 = x1 >> hm_sr_shift(is), sy0 = y0 >> hm_sr_shift(is), sy1 = y1 >> hm_sr_shift(is);
    int local[4];                       // a face larger than 2x2 super-regions reserves its further bins one by one
    int nloc = 0;
    if (mask)
        for (int sy = sy0; sy <= sy1; ++sy)
            for (int sx = sx0; sx <= sx1; ++sx) {
                if (nloc < 4) local[nloc] = atomicAdd(&s_cnt[sy * nsx + sx], 1);
                ++nloc;
            }
    __syncthreads();
    if (threadIdx.x < nsr) {
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c ? atomicAdd(&bin_cnt[b * nsr + threadIdx.x], c) : 0;
    }
    __syncthreads();
    if (mask) {
        int q = 0;
        for (int sy = sy0; sy <= sy1; ++sy)
            for (int sx = sx0; sx <= sx1; ++sx, ++q) {
                const int sr = sy * nsx + sx;
                const int at = q < 4 ? s_base[sr] + local[q] : atomicAdd(&bin_cnt[b * nsr + sr], 1);
                bin_list[((long)b * nsr + sr) * F + at] = fi;
            }
    }
    HM_STAMP_END(0);
}

__global__ __launch_bounds__(256) void k_setup_faces(SetupFacesK a)
{
    setup_faces_body((int)blockIdx.x, (int)blockIdx.y, a.verts, a.K, a.orig_size, a.faces, a.faces_bstride, a.B, a.V, a.F, a.is, a.faces9,
                     a.boxes, a.owned, a.bin_cnt, a.bin_list, a.rigid_rot6d, a.rigid_trans, a.rigid_scale, a.rigid_abs, a.clip_len,
                     a.cam_out, a.nfb);
}
// Several renders as one launch (hm_sil_fwd_multi): grid (largest block count of a render, all renders' frames); the frames of
// render g are rows [first[g], first[g + 1]) of the grid, a render's surplus blocks leave at once.
__global__ __launch_bounds__(256) void k_setup_faces_multi(SetupFacesMulti m)
{
    int g = 0;
#pragma unroll
    for (int k = 1; k < HM_MAX_RENDERS; ++k) g += (k < m.n && (int)blockIdx.y >= m.first[k]) ? 1 : 0;
    g = __builtin_amdgcn_readfirstlane(g);
    const SetupFacesK& a = m.r[g];
    if ((int)blockIdx.x >= m.nblk[g]) return;
    setup_faces_body((int)blockIdx.x, (int)blockIdx.y - m.first[g], a.verts, a.K, a.orig_size, a.faces, a.faces_bstride, a.B, a.V, a.F,
                     a.is, a.faces9, a.boxes, a.owned, a.bin_cnt, a.bin_list, a.rigid_rot6d, a.rigid_trans, a.rigid_scale,
                     a.rigid_abs, a.clip_len, a.cam_out, a.nfb);
}

static SetupFacesK setup_faces_k(const SilWs& w, const SetupFacesArgs& a)
{
    SetupFacesK k = {a.verts, a.K, a.orig_size, a.faces, a.faces_bstride, a.B, a.V, a.F, a.is, w.faces9, w.boxes, w.owned, a.bins,
                     w.bin_list, a.rigid_rot6d, a.rigid_trans, a.rigid_scale, a.rigid_abs, a.clip_len, a.cam_verts_out, hm_cdiv(a.F, 256)};
    return k;
}
void hm_launch_setup_faces(const SilWs& w, const float* verts, const float* K, float orig_size, const int* faces,
                           int faces_bstride, int B, int V, int F, int is, int* bins, const float* rigid_rot6d,
                           const float* rigid_trans, const float* rigid_scale, int rigid_abs, int clip_len,
                           float* cam_verts_out, hipStream_t stream)
{
    const SetupFacesArgs a = {verts, K, orig_size, faces, faces_bstride, B, V, F, is, bins, rigid_rot6d, rigid_trans, rigid_scale,
                              rigid_abs, clip_len, cam_verts_out};
    const int nfb = hm_cdiv(F, 256);
    hipLaunchKernelGGL(k_setup_faces, dim3(nfb + (cam_verts_out ? hm_cdiv(V, 256) : 0), B), dim3(256), 0, stream, setup_faces_k(w, a));
}
void hm_launch_setup_faces_multi(const SilWs* w, const SetupFacesArgs* a, int n, hipStream_t stream)
{
    SetupFacesMulti m;
    m.n = n;
    int rows = 0, cols = 0;
    for (int g = 0; g < HM_MAX_RENDERS; ++g) {
        const int q = g < n ? g : n - 1;           // (unused slots repeat the last render: never selected)
        m.r[g] = setup_faces_k(w[q], a[q]);
        m.first[g] = rows;
        m.nblk[g] = hm_cdiv(a[q].F, 256) + (a[q].cam_verts_out ? hm_cdiv(a[q].V, 256) : 0);
        if (g < n) { rows += a[g].B; cols = max(cols, m.nblk[g]); }
    }
    m.first[HM_MAX_RENDERS] = rows;
    hipLaunchKernelGGL(k_setup_faces_multi, dim3(cols, rows), dim3(256), 0, stream, m);
}

#ifdef HM_CHAIN_STAMPS
extern "C" int hm_debug_chain_raster(unsigned long long* out, int reset)
{
    unsigned long long z[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    (void)hipDeviceSynchronize();
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), z, sizeof(z));
    return HM_OK;
}
#endif
