// contact.hip -- nearest-vertex search (Chamfer direction hand -> object) and the tanh-saturated contact loss.
//
// Replaces reference homan/interactions/contactloss.py:60-79 (batch_pairwise_dist via 3 bmm) + :162-163 (min over
// the object axis) + :11-19 (gather of the nearest object vertex) + the 'dist_tanh' branch (:228-257) and the masked
// means (:50-57,284-285).  As executed by the reference the attraction mask is empty and the
// repulsion mask full, so loss_contact = mean_{b,i} 0.02 * tanh(|nn_i - h_i| / 0.02); gradients flow to the hand
// vertex and to the gathered object vertex.  Also yields the no-grad metric of homan/losses.py:225-241
// (max over frames of the minimum hand-object vertex distance).
// The (B,778,V_o) distance matrix (140 MB at cfg3) is never materialised: object vertices stream through LDS.
#include "hm_common.h"

#define NN_THREADS 256

// grid (ceil(Vh/64), B).  Workgroup = 64 hand vertices x 4 wavefronts; each wave scans one quarter of every
// 1024-vertex object tile staged in LDS (all lanes of a wave read the same object vertex: LDS broadcast), then the
// four partial minima are merged lexicographically on (distance, index) so ties keep the lowest index.
#define NN_HV 64
#define NN_TILE 1024
__global__ __launch_bounds__(NN_THREADS) void k_nn(const float* __restrict__ vh, const float* __restrict__ vo, int B,
                                                    int Vh, int Vo, int* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                                    float* __restrict__ blockmin, unsigned int* counter,
                                                    float* __restrict__ metric_out)
{
    __shared__ float tile[NN_TILE * 3];
    __shared__ float s_d[4][NN_HV];
    __shared__ int s_i[4][NN_HV];
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * NN_HV + lane;
    float hx = 0.f, hy = 0.f, hz = 0.f;
    if (i < Vh) { const float* p = vh + ((long)b * Vh + i) * 3; hx = p[0]; hy = p[1]; hz = p[2]; }
    float best = 3.4e38f;
    int besti = 0;
    for (int j0 = 0; j0 < Vo; j0 += NN_TILE) {
        const int n = min(NN_TILE, Vo - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * n; e += NN_THREADS) tile[e] = vo[((long)b * Vo + j0) * 3 + e];
        __syncthreads();
        const int lo = q * (NN_TILE / 4), hi = min(lo + NN_TILE / 4, n);
        for (int j = lo; j < hi; ++j) {
            const float dx = tile[3 * j] - hx, dy = tile[3 * j + 1] - hy, dz = tile[3 * j + 2] - hz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; besti = j0 + j; }
        }
    }
    s_d[q][lane] = best;
    s_i[q][lane] = besti;
    __syncthreads();
    float bm = 3.4e38f;
    if (q == 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float d = s_d[k][lane];
            const int id = s_i[k][lane];
            if (d < best || (d == best && id < besti)) { best = d; besti = id; }
        }
        if (i < Vh) { nn_idx[(long)b * Vh + i] = besti; nn_d2[(long)b * Vh + i] = best; bm = best; }
    }
    bm = hm_block_min(bm, red);
    const unsigned nblk = gridDim.x * gridDim.y;
    if (threadIdx.x == 0) blockmin[b * gridDim.x + blockIdx.x] = bm;
    if (hm_last_block(counter, nblk, &s_flag)) {
        float mx = -3.4e38f;
        for (int bb = threadIdx.x; bb < B; bb += blockDim.x) {
            float m = 3.4e38f;
            for (unsigned c = 0; c < gridDim.x; ++c) m = fminf(m, blockmin[bb * gridDim.x + c]);
            mx = fmaxf(mx, sqrtf(m));
        }
        mx = hm_block_max(mx, red);
        if (threadIdx.x == 0) metric_out[0] = mx;
    }
}

// grid (1 + ceil(Vo/256), B).  Block x == 0 of a frame: hand side (value, d/d hand vertex, per-frame partial sum,
// last-block finish).  Blocks x >= 1: 256 object vertices each; every thread scans the frame's nearest-neighbour list
// (LDS, broadcast reads) and accumulates, in hand-vertex order, the pulls of the hand vertices that picked it --
// a deterministic scatter without atomics.   loss = mean_{b,i} thresh * tanh(a / thresh).
__device__ __forceinline__ float contact_pull(const float* h, const float* o, float thresh, float inv_cnt, float* pull,
                                              float* value)
{
    const float dx = o[0] - h[0], dy = o[1] - h[1], dz = o[2] - h[2];
    const float a = sqrtf(dx * dx + dy * dy + dz * dz);
    const float th = tanhf(a / thresh);
    const float k = (a > 0.f) ? (1.0f - th * th) / a * inv_cnt : 0.f;     // d val / d a  / a
    pull[0] = k * dx; pull[1] = k * dy; pull[2] = k * dz;                  // d/d o ; d/d h is the negative
    if (value) *value = thresh * th;
    return k;
}

__global__ __launch_bounds__(NN_THREADS) void k_contact(const float* __restrict__ vh, const float* __restrict__ vo,
                                                         const int* __restrict__ nn_idx, int B, int Vh, int Vo,
                                                         float thresh, float* __restrict__ g_hand,
                                                         float* __restrict__ g_obj, float* __restrict__ partials,
                                                         unsigned int* counter, float* __restrict__ out)
{
    extern __shared__ int s_idx[];          // Vh ints (object blocks only)
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = blockIdx.y;
    const float inv_cnt = 1.0f / (float)((long)B * Vh);
    if (blockIdx.x == 0) {
        float lsum = 0.f;
        for (int i = threadIdx.x; i < Vh; i += NN_THREADS) {
            const int j = nn_idx[(long)b * Vh + i];
            float pull[3], val;
            contact_pull(vh + ((long)b * Vh + i) * 3, vo + ((long)b * Vo + j) * 3, thresh, inv_cnt, pull, &val);
            lsum += val;
            float* gh = g_hand + ((long)b * Vh + i) * 3;
            gh[0] = -pull[0]; gh[1] = -pull[1]; gh[2] = -pull[2];
        }
        lsum = hm_block_sum(lsum, red);
        if (threadIdx.x == 0) partials[b] = lsum;
        if (hm_last_block(counter, gridDim.y, &s_flag)) {
            const float t = hm_last_block_sum(partials, B, 1, red);
            if (threadIdx.x == 0) out[0] = t * inv_cnt;
        }
        return;
    }
    for (int i = threadIdx.x; i < Vh; i += NN_THREADS) s_idx[i] = nn_idx[(long)b * Vh + i];
    __syncthreads();
    const int j = (blockIdx.x - 1) * NN_THREADS + threadIdx.x;
    if (j >= Vo) return;
    const float* o = vo + ((long)b * Vo + j) * 3;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int i = 0; i < Vh; ++i)
        if (s_idx[i] == j) {
            float pull[3];
            contact_pull(vh + ((long)b * Vh + i) * 3, o, thresh, inv_cnt, pull, nullptr);
            gx += pull[0]; gy += pull[1]; gz += pull[2];
        }
    float* go = g_obj + ((long)b * Vo + j) * 3;
    go[0] = gx; go[1] = gy; go[2] = gz;
}

extern "C" {
// workspace: reuse hm_reduce_workspace_bytes() layout (partials + counter); needs B <= 512 and
// B*ceil(Vh/256) <= 512 partial floats.
int hm_nn_fwd(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
              float* metric_out, void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts_hand && verts_obj && nn_idx && nn_d2 && metric_out && workspace && B > 0 && Vh > 0 && Vo > 0);
    const int nchunk = hm_cdiv(Vh, NN_HV);
    if ((long)B * nchunk > 512) return HM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_nn, dim3(nchunk, B), dim3(NN_THREADS), 0, stream, verts_hand, verts_obj, B, Vh, Vo, nn_idx,
                       nn_d2, (float*)workspace, (unsigned int*)((float*)workspace + 512), metric_out);
    return hm_launch_status();
}
int hm_contact_fwd(const float* verts_hand, const float* verts_obj, const int* nn_idx, int B, int Vh, int Vo,
                   float thresh, float* g_hand, float* g_obj, float* out1, void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts_hand && verts_obj && nn_idx && g_hand && g_obj && out1 && workspace);
    HM_CHECK_ARG(B > 0 && B <= 512 && Vh > 0 && Vo > 0 && (size_t)Vh * 16 <= 60000);
    hipLaunchKernelGGL(k_contact, dim3(1 + hm_cdiv(Vo, NN_THREADS), B), dim3(NN_THREADS), (size_t)Vh * sizeof(int), stream, verts_hand, verts_obj,
                       nn_idx, B, Vh, Vo, thresh, g_hand, g_obj, (float*)workspace,
                       (unsigned int*)((float*)workspace + 512), out1);
    return hm_launch_status();
}
}  // extern "C"
