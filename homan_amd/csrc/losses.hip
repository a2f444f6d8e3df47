// losses.hip -- the small per-vertex / per-frame losses of the reference, forward value + unit gradient
// in one launch each (the backward is then a scalar scaling).
//
//   hm_v2d_fwd        reference homan/losses.py:141-164  (2-D reprojection of hand vertices + px metric)
//   hm_smooth_fwd     reference homan/lossutils.py:18-36 (temporal smoothness, hand or object)
//   hm_priors_fwd     reference homan/lossutils.py:39-40 (PCA prior) and :107-109 (intrinsic scale priors)
//   hm_inter_fwd/bwd  reference homan/losses.py:20-49,98-139,199-242 + utils/bbox.py:111-135 +
//                     utils/geometry.py:69-86 (coarse interaction: bbox-IoU / z gating, centroid MSE, summed)
#include "hm_common.h"
#include "pair_bodies.h"

// ------------------------------------------------------------------ clip batches
// Every reduction kernel below takes grid (nblk, clips): row blockIdx.y works on clip blockIdx.y - N frames of its own,
// its own slice of the reduce workspace (HM_RED_WS_FLOATS floats: partial records + ticket), its own output row - exactly
// as a single-clip launch with grid (nblk) would.  CLIP_ADVANCE moves a per-frame pointer to the clip's first frame.
#define CLIP_ADVANCE(ptr, per_frame) ptr += (long)blockIdx.y * N * (per_frame)
#define CLIP_WS(partials, counter) partials += (long)blockIdx.y * HM_RED_WS_FLOATS; counter += (long)blockIdx.y * HM_RED_WS_FLOATS

// ------------------------------------------------------------------ v2d
// grid (nblk, clips). partial records: 2 floats per block.
__global__ __launch_bounds__(RED_THREADS) void k_v2d(const float* __restrict__ verts, const float* __restrict__ camintr,
                                                      int hand_nb, const float* __restrict__ ref2d, float image_size,
                                                      int N, int V, float* __restrict__ unit_grad,
                                                      float* __restrict__ partials, unsigned int* counter,
                                                      float* __restrict__ out, int out_stride)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    CLIP_ADVANCE(verts, V * 3); CLIP_ADVANCE(ref2d, V * 2); CLIP_ADVANCE(unit_grad, V * 3);
    camintr += (long)blockIdx.y * (N / hand_nb) * 9;
    CLIP_WS(partials, counter);
    out += (long)blockIdx.y * out_stride;
    const long total = (long)N * V;
    const float inv_cnt = 1.0f / (float)total;
    float lsum = 0.f, msum = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / V);
        const float* k = camintr + (n / hand_nb) * 9;
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        const float hx = k[0] * x + k[1] * y + k[2] * z;
        const float hy = k[3] * x + k[4] * y + k[5] * z;
        const float hz = k[6] * x + k[7] * y + k[8] * z;
        const float px = hx / hz, py = hy / hz;
        const float rx = ref2d[2 * i], ry = ref2d[2 * i + 1];
        const float dx = px - rx / image_size, dy = py - ry / image_size;
        lsum += dx * dx + dy * dy;
        const float mx = px * image_size - rx, my = py * image_size - ry;
        msum += sqrtf(mx * mx + my * my);
        // d loss / d (px,py) = 2 d / count
        const float gpx = 2.0f * dx * inv_cnt, gpy = 2.0f * dy * inv_cnt;
        const float ghx = gpx / hz, ghy = gpy / hz, ghz = -(gpx * hx + gpy * hy) / (hz * hz);
        unit_grad[3 * i] = k[0] * ghx + k[3] * ghy + k[6] * ghz;
        unit_grad[3 * i + 1] = k[1] * ghx + k[4] * ghy + k[7] * ghz;
        unit_grad[3 * i + 2] = k[2] * ghx + k[5] * ghy + k[8] * ghz;
    }
    lsum = hm_block_sum(lsum, red);
    msum = hm_block_sum(msum, red);
    if (threadIdx.x == 0) { hm_partial_store(partials + 2 * blockIdx.x, lsum); hm_partial_store(partials + 2 * blockIdx.x + 1, msum); }
    if (hm_last_block(counter, gridDim.x, &s_flag)) {
        const float a = hm_last_block_sum(partials, gridDim.x, 2, red);
        const float b = hm_last_block_sum(partials + 1, gridDim.x, 2, red);
        if (threadIdx.x == 0) { out[0] = a * inv_cnt; out[1] = b * inv_cnt; }
    }
}

// ------------------------------------------------------------------ temporal smoothness
// verts (N,V,3), frames interleaved by `hand_nb` (pairs n, n+hand_nb).  loss = mean over pairs of diff^2.
__global__ __launch_bounds__(RED_THREADS) void k_smooth(const float* __restrict__ verts, int N, int V, int hand_nb,
                                                         float* __restrict__ unit_grad, float* __restrict__ partials,
                                                         unsigned int* counter, float* __restrict__ out, int out_stride)
{
    smooth_body(verts, N, V, hand_nb, unit_grad, partials, counter, out, out_stride, blockIdx.x, blockIdx.y, gridDim.x);
}

// ------------------------------------------------------------------ PCA prior + intrinsic scale priors (one block per clip)
// out[0] = mean(pca^2), out[1] = sum((s_obj - m_obj)^2)/1, out[2] = same for the hand scale.  grid (1, clips); npca =
// PCA entries of ONE clip; the scales / means / scale gradients are per-clip arrays.
__global__ __launch_bounds__(RED_THREADS) void k_priors(const float* __restrict__ pca, long npca,
                                                         const float* __restrict__ s_obj, const float* __restrict__ m_obj,
                                                         const float* __restrict__ s_hand, const float* __restrict__ m_hand,
                                                         float* __restrict__ g_pca, float* __restrict__ g_sobj,
                                                         float* __restrict__ g_shand, float* __restrict__ out,
                                                         int out_stride)
{
    __shared__ float red[16];
    pca += (long)blockIdx.y * npca; g_pca += (long)blockIdx.y * npca;
    s_obj += blockIdx.y; m_obj += blockIdx.y; s_hand += blockIdx.y; m_hand += blockIdx.y;
    g_sobj += blockIdx.y; g_shand += blockIdx.y;
    out += (long)blockIdx.y * out_stride;
    float a = 0.f;
    const float inv = 1.0f / (float)npca;
    for (long i = threadIdx.x; i < npca; i += blockDim.x) {
        const float p = pca[i];
        a += p * p;
        g_pca[i] = 2.0f * p * inv;
    }
    a = hm_block_sum(a, red);
    if (threadIdx.x == 0) {
        out[0] = a * inv;
        const float d0 = s_obj[0] - m_obj[0], d1 = s_hand[0] - m_hand[0];
        out[1] = d0 * d0;
        out[2] = d1 * d1;
        g_sobj[0] = 2.0f * d0;
        g_shand[0] = 2.0f * d1;
    }
}

// ------------------------------------------------------------------ hand terms in one launch
// The 2-D reprojection term, the temporal smoothness term (both over the same hand vertices) and the priors are three
// small reductions that used to be three launches on the hand-side critical chain; here they share one grid and one
// "last block finishes" ticket.  Same per-element arithmetic as k_v2d / k_smooth / k_priors; partial records: 3 floats
// per block (v2d loss, px metric, smoothness).  pca == NULL skips the priors.
__global__ __launch_bounds__(RED_THREADS) void k_hand_terms(
    const float* __restrict__ verts, const float* __restrict__ camintr, int hand_nb, const float* __restrict__ ref2d,
    float image_size, int N, int V, float* __restrict__ unit_v2d, float* __restrict__ out_v2d,
    float* __restrict__ unit_smooth, float* __restrict__ out_smooth, const float* __restrict__ pca, long npca,
    const float* __restrict__ s_obj, const float* __restrict__ m_obj, const float* __restrict__ s_hand,
    const float* __restrict__ m_hand, float* __restrict__ g_pca, float* __restrict__ g_sobj,
    float* __restrict__ g_shand, float* __restrict__ out_priors, float* __restrict__ partials, unsigned int* counter,
    int out_stride)
{
    hand_terms_body(verts, camintr, hand_nb, ref2d, image_size, N, V, unit_v2d, out_v2d, unit_smooth, out_smooth, pca, npca, s_obj,
                    m_obj, s_hand, m_hand, g_pca, g_sobj, g_shand, out_priors, partials, counter, out_stride, blockIdx.x,
                    blockIdx.y, gridDim.x);
}

// ------------------------------------------------------------------ coarse interaction loss
// grid (B).  Per frame: expanded 2-D boxes of the projected meshes (y negated, nr.projection with the
// normalised camera, orig_size 1), IoU>0 and z-gap<thresh gate, MSE of the two centroids.
// frame record (8 floats): flag, mse, gvec[3] (= flag*2*(ch-co)/3), pad.
__global__ __launch_bounds__(RED_THREADS) void k_inter(const float* __restrict__ vh, const float* __restrict__ vo,
                                                        const float* __restrict__ camintr, int B, int Vh, int Vo,
                                                        float expansion, float zthresh, float* __restrict__ frame_rec,
                                                        unsigned int* counter, float* __restrict__ out, int clip_len,
                                                        int out_stride)
{
    HM_LATENCY_KERNEL();
    inter_body(vh, vo, camintr, B, Vh, Vo, expansion, zthresh, frame_rec, counter, out, clip_len, out_stride, blockIdx.x);
}

// d loss_inter / d verts: hand gets +gvec/Vh, object gets -gvec/Vo (either output may be NULL)
__global__ void k_inter_bwd(const float* __restrict__ frame_rec, const float* __restrict__ upstream, int B, int Vh,
                            int Vo, float* __restrict__ g_hand, float* __restrict__ g_obj)
{
    HM_LATENCY_KERNEL();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nh = (long)B * Vh * 3, no = (long)B * Vo * 3;
    const float up = upstream[0];
    if (g_hand && i < nh) {
        const int b = (int)(i / ((long)Vh * 3)), c = (int)(i % 3);
        g_hand[i] = up * frame_rec[b * 8 + 2 + c] / (float)Vh;
    }
    if (g_obj && i < no) {
        const int b = (int)(i / ((long)Vo * 3)), c = (int)(i % 3);
        g_obj[i] = -up * frame_rec[b * 8 + 2 + c] / (float)Vo;
    }
}

// ------------------------------------------------------------------ off-screen penalty of the pose initialisation
// reference homan/pose_optimization.py:112-135: hinge on the six clipping planes of the projected vertices (NDC x, y in
// [-1,1] under nr.projection(K, R=I, t=0, orig_size=1), 0 < z < far), summed over the vertices of one candidate pose.
// grid (N): value out[n] = weight * sum, and d out / d vertex (N,V,3) in the same pass (K: ONE 3x3 camera for all poses).
__global__ __launch_bounds__(256) void k_offscreen(const float* __restrict__ verts, const float* __restrict__ K, int V,
                                                   float zfar, float weight, float* __restrict__ out,
                                                   float* __restrict__ grad)
{
    __shared__ float red[16];
    const int n = blockIdx.x;
    const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5];
    float acc = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const long o = ((long)n * V + v) * 3;
        const float x = verts[o], y = verts[o + 1], z = verts[o + 2];
        const float zz = z + 1e-9f;
        const float xn = x / zz, yn = y / zz;
        float u = k00 * xn + k01 * yn;
        u = u + k02;
        float w = k10 * xn + k11 * yn;
        w = 1.0f - (w + k12);
        const float nu = 2.0f * (u - 0.5f), nv = 2.0f * (w - 0.5f);
        // relu(ndc - 1) + relu(-1 - ndc) per component, relu(-z), relu(z - far); d relu / dx = [x > 0]
        float val = fmaxf(nu - 1.0f, 0.f) + fmaxf(nv - 1.0f, 0.f);
        val += fmaxf(-1.0f - nu, 0.f) + fmaxf(-1.0f - nv, 0.f);
        val += fmaxf(-z, 0.f);
        val += fmaxf(z - zfar, 0.f);
        acc += val;
        const float gu = (nu - 1.0f > 0.f ? 1.f : 0.f) - (-1.0f - nu > 0.f ? 1.f : 0.f);
        const float gv = (nv - 1.0f > 0.f ? 1.f : 0.f) - (-1.0f - nv > 0.f ? 1.f : 0.f);
        const float gz = (z - zfar > 0.f ? 1.f : 0.f) - (-z > 0.f ? 1.f : 0.f);
        const float du = 2.0f * gu, dw = -2.0f * gv;             // d / d u, d / d (k10 xn + k11 yn + k12)
        const float dxn = k00 * du + k10 * dw, dyn = k01 * du + k11 * dw;
        grad[o] = weight * (dxn / zz);
        grad[o + 1] = weight * (dyn / zz);
        grad[o + 2] = weight * (gz - (dxn * x + dyn * y) / (zz * zz));
    }
    acc = hm_block_sum(acc, red);
    if (threadIdx.x == 0) out[n] = weight * acc;
}

// Best-ever bookkeeping of the pose initialisation's loop (reference homan/pose_optimization.py:340-353), one workgroup:
// losses[i] = sums[i * stride] + extra[i]; (lmin, ind) = first minimum (torch.argmin); if lmin < best_loss[0] (strict; a NaN
// among the losses makes the minimum NaN, as torch.min does: no update) the pose of candidate `ind` - as it is NOW, i.e. after
// the optimiser step that followed the evaluation - becomes the best one.
// log != NULL (hm_pose_keep_best_log): no best-ever state is touched; the step's record - {minimum, its candidate (int bits), "some
// loss is NaN", that candidate's rot6d (6) and trans (3), 0 x 4} - goes to row step[0] - 1 of `log` (step[0] = the optimiser's step
// counter AFTER the step that followed the evaluation), for a caller that walks the candidates as several independent loops and
// applies the rule below over all of them afterwards.
__global__ __launch_bounds__(256) void k_pose_keep_best(const float* __restrict__ sums, int stride, const float* __restrict__ extra,
                                                        int n, const float* __restrict__ rot6d, const float* __restrict__ trans,
                                                        float* __restrict__ best_loss, float* __restrict__ best_rot,
                                                        float* __restrict__ best_trans, float* __restrict__ losses_out,
                                                        const int* __restrict__ step, int max_steps, float* __restrict__ log)
{
    __shared__ float s_v[4];
    __shared__ int s_i[4], s_nan[4];
    float v = INFINITY;
    int at = 0x7fffffff, bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float l = sums[(long)i * stride] + extra[i];
        losses_out[i] = l;
        if (l != l) bad = 1;
        if (l < v) { v = l; at = i; }           // (ascending i per thread: the first minimum of the thread)
    }
    // wave minimum with the lowest index among equals, then the four waves
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(at, o);
        bad |= __shfl_xor(bad, o);
        if (ov < v || (ov == v && oi < at)) { v = ov; at = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_v[w] = v; s_i[w] = at; s_nan[w] = bad; }
    __syncthreads();
    if (threadIdx.x < 9) {
        float bv = s_v[0];
        int bi = s_i[0], nb = s_nan[0];
        for (int k = 1; k < 4; ++k) {
            nb |= s_nan[k];
            if (s_v[k] < bv || (s_v[k] == bv && s_i[k] < bi)) { bv = s_v[k]; bi = s_i[k]; }
        }
        if (log) {
            float* rec = log + (long)min(max(step[0] - 1, 0), max_steps - 1) * 16;
            const int e = threadIdx.x;
            const bool has = bi < n;
            if (e < 6) rec[3 + e] = has ? rot6d[(long)bi * 6 + e] : 0.f;
            else rec[3 + e] = has ? trans[(long)bi * 3 + (e - 6)] : 0.f;
            if (e == 0) { rec[0] = bv; rec[1] = __int_as_float(has ? bi : -1); rec[2] = nb ? 1.f : 0.f; }
            if (e < 4) rec[12 + e] = 0.f;
            return;
        }
        // (lanes 0..8 are one wave: all of them have read best_loss before lane 0 writes it)
        const bool better = !nb && bi < n && bv < best_loss[0];
        if (better) {
            const int e = threadIdx.x;
            if (e < 6) best_rot[e] = rot6d[(long)bi * 6 + e];
            else best_trans[e - 6] = trans[(long)bi * 3 + (e - 6)];
            if (e == 0) best_loss[0] = bv;
        }
    }
}

extern "C" {
int hm_pose_keep_best(const float* sums, int stride, const float* extra, int n, const float* rot6d, const float* trans,
                      float* best_loss, float* best_rot6d, float* best_trans, float* losses_out, hipStream_t stream)
{
    HM_CHECK_ARG(sums && extra && rot6d && trans && best_loss && best_rot6d && best_trans && losses_out && n > 0 && stride > 0);
    hipLaunchKernelGGL(k_pose_keep_best, dim3(1), dim3(256), 0, stream, sums, stride, extra, n, rot6d, trans, best_loss,
                       best_rot6d, best_trans, losses_out, (const int*)nullptr, 0, (float*)nullptr);
    return hm_launch_status();
}
int hm_pose_keep_best_log(const float* sums, int stride, const float* extra, int n, const float* rot6d, const float* trans,
                          const int* step, int max_steps, float* log, float* losses_out, hipStream_t stream)
{
    HM_CHECK_ARG(sums && extra && rot6d && trans && step && log && losses_out && n > 0 && stride > 0 && max_steps > 0);
    hipLaunchKernelGGL(k_pose_keep_best, dim3(1), dim3(256), 0, stream, sums, stride, extra, n, rot6d, trans, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, losses_out, step, max_steps, log);
    return hm_launch_status();
}
int hm_offscreen_fwd(const float* verts, const float* K, int N, int V, float zfar, float weight, float* out, float* grad,
                     hipStream_t stream)
{
    HM_CHECK_ARG(verts && K && out && grad && N > 0 && V > 0);
    hipLaunchKernelGGL(k_offscreen, dim3(N), dim3(256), 0, stream, verts, K, V, zfar, weight, out, grad);
    return hm_launch_status();
}
#define HM_RED_MAX_BLOCKS 256
// workspace for the reductions: 512 floats of partials + 1 counter word at float offset 512 (HM_RED_WS_FLOATS floats in
// all; the whole buffer must be zero-initialised once; the counter resets itself).  A *_clips call needs one such slice
// per clip, back to back.
size_t hm_reduce_workspace_bytes(void) { return HM_RED_WS_FLOATS * sizeof(float); }

static inline unsigned int* ws_counter(void* ws) { return (unsigned int*)((float*)ws + 512); }
// N frames in clips of clip_len (0: one clip) -> frames per clip, number of clips
static inline int clip_frames(int N, int clip_len) { return clip_len ? clip_len : N; }
static inline int clip_count(int N, int clip_len) { return clip_len ? N / clip_len : 1; }

int hm_v2d_fwd_clips(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
                     int V, float* unit_grad, float* out2, void* workspace, int clip_len, int out_stride,
                     hipStream_t stream)
{
    HM_CHECK_ARG(verts && camintr && ref2d && unit_grad && out2 && workspace && N > 0 && V > 0 && hand_nb > 0);
    HM_CHECK_ARG(HM_CLIP_LEN_OK(N, clip_len));
    const int Nc = clip_frames(N, clip_len);
    const int nblk = min(HM_RED_MAX_BLOCKS, hm_cdiv((long)Nc * V, RED_THREADS));
    hipLaunchKernelGGL(k_v2d, dim3(nblk, clip_count(N, clip_len)), dim3(RED_THREADS), 0, stream, verts, camintr, hand_nb,
                       ref2d, image_size, Nc, V, unit_grad, (float*)workspace, ws_counter(workspace), out2, out_stride);
    return hm_launch_status();
}
int hm_v2d_fwd(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
               int V, float* unit_grad, float* out2, void* workspace, hipStream_t stream)
{
    return hm_v2d_fwd_clips(verts, camintr, hand_nb, ref2d, image_size, N, V, unit_grad, out2, workspace, 0, 0, stream);
}
int hm_smooth_fwd_clips(const float* verts, int N, int V, int hand_nb, float* unit_grad, float* out1, void* workspace,
                        int clip_len, int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(verts && unit_grad && out1 && workspace && N > 0 && V > 0 && hand_nb > 0 && HM_CLIP_LEN_OK(N, clip_len));
    const int Nc = clip_frames(N, clip_len);
    const int nblk = min(HM_RED_MAX_BLOCKS, hm_cdiv((long)Nc * V * 3, RED_THREADS * 4));
    hipLaunchKernelGGL(k_smooth, dim3(nblk, clip_count(N, clip_len)), dim3(RED_THREADS), g_hm_lds_pad[HM_PAD_SMALL_LOSSES], stream, verts, Nc, V, hand_nb,
                       unit_grad, (float*)workspace, ws_counter(workspace), out1, out_stride);
    return hm_launch_status();
}
int hm_smooth_fwd(const float* verts, int N, int V, int hand_nb, float* unit_grad, float* out1, void* workspace,
                  hipStream_t stream)
{
    return hm_smooth_fwd_clips(verts, N, V, hand_nb, unit_grad, out1, workspace, 0, 0, stream);
}
// npca: PCA entries of ONE clip; the scales, their means and the scale gradients hold one entry per clip.
int hm_priors_fwd_clips(const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                        const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out3, int nclips,
                        int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(pca && s_obj && m_obj && s_hand && m_hand && g_pca && g_sobj && g_shand && out3 && npca > 0 && nclips > 0);
    hipLaunchKernelGGL(k_priors, dim3(1, nclips), dim3(RED_THREADS), 0, stream, pca, npca, s_obj, m_obj, s_hand, m_hand,
                       g_pca, g_sobj, g_shand, out3, out_stride);
    return hm_launch_status();
}
int hm_priors_fwd(const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                  const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out3, hipStream_t stream)
{
    return hm_priors_fwd_clips(pca, npca, s_obj, m_obj, s_hand, m_hand, g_pca, g_sobj, g_shand, out3, 1, 0, stream);
}
// v2d + smoothness (+ priors when pca != NULL) of the hand vertices in one launch: same outputs as hm_v2d_fwd,
// hm_smooth_fwd and hm_priors_fwd called one after the other.  (npca: PCA entries of ONE clip.)
int hm_hand_terms_fwd_clips(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size,
                            int N, int V, float* unit_v2d, float* out_v2d2, float* unit_smooth, float* out_smooth1,
                            const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                            const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out_priors3,
                            void* workspace, int clip_len, int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(verts && camintr && ref2d && unit_v2d && out_v2d2 && unit_smooth && out_smooth1 && workspace);
    HM_CHECK_ARG(N > 0 && V > 0 && hand_nb > 0 && HM_CLIP_LEN_OK(N, clip_len));
    HM_CHECK_ARG(!pca || (npca > 0 && s_obj && m_obj && s_hand && m_hand && g_pca && g_sobj && g_shand && out_priors3));
    const int Nc = clip_frames(N, clip_len);
    const int nblk = min(170, hm_cdiv((long)Nc * V * 3, RED_THREADS * 2));     // 3 partial floats per block, 512 in all
    hipLaunchKernelGGL(k_hand_terms, dim3(nblk, clip_count(N, clip_len)), dim3(RED_THREADS), g_hm_lds_pad[HM_PAD_SMALL_LOSSES], stream, verts, camintr,
                       hand_nb, ref2d, image_size, Nc, V, unit_v2d, out_v2d2, unit_smooth, out_smooth1, pca, npca, s_obj,
                       m_obj, s_hand, m_hand, g_pca, g_sobj, g_shand, out_priors3, (float*)workspace,
                       ws_counter(workspace), out_stride);
    return hm_launch_status();
}
int hm_hand_terms_fwd(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
                      int V, float* unit_v2d, float* out_v2d2, float* unit_smooth, float* out_smooth1, const float* pca,
                      long npca, const float* s_obj, const float* m_obj, const float* s_hand, const float* m_hand,
                      float* g_pca, float* g_sobj, float* g_shand, float* out_priors3, void* workspace, hipStream_t stream)
{
    return hm_hand_terms_fwd_clips(verts, camintr, hand_nb, ref2d, image_size, N, V, unit_v2d, out_v2d2, unit_smooth,
                                   out_smooth1, pca, npca, s_obj, m_obj, s_hand, m_hand, g_pca, g_sobj, g_shand,
                                   out_priors3, workspace, 0, 0, stream);
}
// frame_rec: (B,8) floats kept for the backward.
int hm_inter_fwd_clips(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                       float expansion, float zthresh, float* frame_rec, float* out1, void* workspace, int clip_len,
                       int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(verts_hand && verts_obj && camintr && frame_rec && out1 && workspace && B > 0 && Vh > 0 && Vo > 0);
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    hipLaunchKernelGGL(k_inter, dim3(B), dim3(RED_THREADS), g_hm_lds_pad[HM_PAD_SMALL_LOSSES], stream, verts_hand, verts_obj, camintr, B, Vh, Vo,
                       expansion, zthresh, frame_rec, ws_counter(workspace), out1, clip_frames(B, clip_len), out_stride);
    return hm_launch_status();
}
int hm_inter_fwd(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                 float expansion, float zthresh, float* frame_rec, float* out1, void* workspace, hipStream_t stream)
{
    return hm_inter_fwd_clips(verts_hand, verts_obj, camintr, B, Vh, Vo, expansion, zthresh, frame_rec, out1, workspace,
                              0, 0, stream);
}
int hm_inter_bwd(const float* frame_rec, const float* upstream, int B, int Vh, int Vo, float* g_hand, float* g_obj,
                 hipStream_t stream)
{
    HM_CHECK_ARG(frame_rec && upstream && B > 0 && (g_hand || g_obj));
    const long n = (long)B * (Vh > Vo ? Vh : Vo) * 3;
    hipLaunchKernelGGL(k_inter_bwd, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, frame_rec, upstream, B, Vh, Vo, g_hand,
                       g_obj);
    return hm_launch_status();
}
}  // extern "C"
