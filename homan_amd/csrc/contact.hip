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
#include "pair_bodies.h"

#define NN_THREADS 256

// grid (ceil(Vh/128), B): nn_full_body (pair_bodies.h) as a launch of its own
__global__ __launch_bounds__(64 * NN_WAVES) void k_nn(const float* __restrict__ vh, const float* __restrict__ vo, int B,
                                                       int Vh, int Vo, int* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                                       float* __restrict__ blockmin, unsigned int* counter,
                                                       float* __restrict__ metric_out, int clip_len, int out_stride)
{
    HM_LATENCY_KERNEL();
    nn_full_body(vh, vo, B, Vh, Vo, nn_idx, nn_d2, blockmin, counter, metric_out, clip_len, out_stride, blockIdx.x, blockIdx.y,
                 gridDim.x);
}

// METRIC ONLY (the step-1 loss sets, where the search feeds nothing but the logged hand-object distance of reference
// losses.py:225-241): only the smallest distance of the frame matters.  grid (ceil(Vh/128), B), 4 waves, all holding the
// same 128 hand vertices (2 per lane).
//   1. bounding spheres of the groups of 64 object vertices (visited in `obj_order`: a spatial sort of the rigid mesh makes
//      a group a compact patch), groups dealt to the waves round-robin;
//   2. per group: the hand vertex nearest to its centre gives an UPPER bound on the frame's minimum (centre distance +
//      radius: some vertex of the group is at least that close) and a LOWER bound for the group (centre distance - radius);
//   3. only the groups whose lower bound does not exceed the best upper bound are scanned exactly (same arithmetic as k_nn):
//      the closest approach of hand and object is a small neighbourhood, typically 2-3 of ~24 groups.
// The result is the exact minimum (bounds carry a 1e-5 margin for their own rounding); blockmin / ticket / finish as k_nn.
__global__ __launch_bounds__(64 * NN_WAVES) void k_nn_min(const float* __restrict__ vh, const float* __restrict__ vo, int B,
                                                           int Vh, int Vo, float* __restrict__ blockmin,
                                                           unsigned int* counter, float* __restrict__ metric_out,
                                                           int clip_len, int out_stride, const int* __restrict__ obj_order,
                                                           const float* __restrict__ sph_mesh,
                                                           const float* __restrict__ obj_rot6d,
                                                           const float* __restrict__ obj_trans,
                                                           const float* __restrict__ obj_scale,
                                                           const int* __restrict__ hand_order, int* __restrict__ seed)
{
    HM_LATENCY_KERNEL();
    nn_min_body(vh, vo, B, Vh, Vo, blockmin, counter, metric_out, clip_len, out_stride, obj_order, blockIdx.x, blockIdx.y,
                gridDim.x, sph_mesh, obj_rot6d, obj_trans, obj_scale, hand_order, seed);
}

// Contact loss, hand side.  grid (B): value, d/d hand vertex (= minus the pull on the matched object vertex),
// per-frame partial sums, last-block finish.   loss = mean_{b,i} thresh * tanh(a / thresh).
__global__ __launch_bounds__(NN_THREADS) void k_contact_hand(const float* __restrict__ vh, const float* __restrict__ vo,
                                                              const int* __restrict__ nn_idx, int B, int Vh, int Vo,
                                                              float thresh, float* __restrict__ g_hand,
                                                              float* __restrict__ partials, unsigned int* counter,
                                                              float* __restrict__ out, int clip_len, int out_stride)
{
    HM_LATENCY_KERNEL();
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = blockIdx.x, clip = b / clip_len, bl = b - clip * clip_len;
    partials += (long)clip * HM_RED_WS_FLOATS;
    counter += (long)clip * HM_RED_WS_FLOATS;
    const float inv_cnt = 1.0f / (float)((long)clip_len * Vh);        // the mean runs over the clip's frames
    float lsum = 0.f;
    for (int i = threadIdx.x; i < Vh; i += NN_THREADS) {
        const int j = nn_idx[(long)b * Vh + i];
        const float* h = vh + ((long)b * Vh + i) * 3;
        const float* o = vo + ((long)b * Vo + j) * 3;
        const float dx = o[0] - h[0], dy = o[1] - h[1], dz = o[2] - h[2];
        const float a = sqrtf(dx * dx + dy * dy + dz * dz);
        const float th = hm_tanh(a / thresh);
        lsum += thresh * th;
        const float k = (a > 0.f) ? (1.0f - th * th) / a * inv_cnt : 0.f;     // d val / d a  / a
        float* gh = g_hand + ((long)b * Vh + i) * 3;                          // d a / d h = -diff / a
        gh[0] = -k * dx; gh[1] = -k * dy; gh[2] = -k * dz;
    }
    lsum = hm_block_sum(lsum, red);
    if (threadIdx.x == 0) hm_partial_store(partials + bl, lsum);
    if (hm_last_block(counter, clip_len, &s_flag)) {
        const float t = hm_last_block_sum(partials, clip_len, 1, red);
        if (threadIdx.x == 0) out[(long)clip * out_stride] = t * inv_cnt;
    }
}

// Contact loss, object side.  grid (B): the gradient of an object vertex is minus the sum of the gradients of the hand
// vertices that picked it.  The picks are heavily skewed (the whole hand usually lands on a few dozen object vertices),
// so a gather per object vertex serialises on the popular ones; instead every hand vertex adds into a per-frame LDS
// accumulator with 64-bit FIXED-POINT atomics (2^-44 units): integer addition is associative, so the result does not
// depend on the order the atomics land in -- deterministic without a sort.  |g| <= 1/(B*Vh) here, far inside the range.
#define CONTACT_MAX_VO 4096                    // object vertices per workgroup (96 KB of LDS accumulators)
#define CONTACT_FIX 17592186044416.0f          // 2^44
// grid (B, ceil(Vo / CONTACT_MAX_VO)): a workgroup owns one range of object vertices and adds the hand vertices that picked
// a vertex of its range (every workgroup of the frame walks the frame's 778 picks; meshes of up to 4096 vertices are one range)
__global__ __launch_bounds__(NN_THREADS) void k_contact_obj(const int* __restrict__ nn_idx, const float* __restrict__ g_hand,
                                                             int B, int Vh, int Vo, float* __restrict__ g_obj)
{
    HM_LATENCY_KERNEL();
    __shared__ unsigned long long acc[CONTACT_MAX_VO * 3];
    const int b = blockIdx.x, lo = blockIdx.y * CONTACT_MAX_VO, n = min(Vo - lo, CONTACT_MAX_VO);
    for (int i = threadIdx.x; i < 3 * n; i += NN_THREADS) acc[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < Vh; i += NN_THREADS) {
        const int j = nn_idx[(long)b * Vh + i] - lo;
        if (j < 0 || j >= n) continue;
        const float* gh = g_hand + ((long)b * Vh + i) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            atomicAdd(&acc[3 * j + c], (unsigned long long)(long long)__float2ll_rn(gh[c] * CONTACT_FIX));
    }
    __syncthreads();
    float* go = g_obj + ((long)b * Vo + lo) * 3;
    for (int i = threadIdx.x; i < 3 * n; i += NN_THREADS)
        go[i] = -(float)(long long)acc[i] * (1.0f / CONTACT_FIX);
}

// Both sides in ONE launch for meshes of one range (Vo <= CONTACT_MAX_VO): the hand vertex's gradient goes to memory and,
// from the register, into the frame's fixed-point accumulators - the same values and the same (order-free) integer sums as
// the two launches, one launch less on the hand-side chain of the step-2 loss sets.  grid (B)
__global__ __launch_bounds__(NN_THREADS) void k_contact_both(const float* __restrict__ vh, const float* __restrict__ vo,
                                                              const int* __restrict__ nn_idx, int B, int Vh, int Vo,
                                                              float thresh, float* __restrict__ g_hand,
                                                              float* __restrict__ g_obj, float* __restrict__ partials,
                                                              unsigned int* counter, float* __restrict__ out, int clip_len,
                                                              int out_stride)
{
    HM_LATENCY_KERNEL();
    __shared__ unsigned long long acc[CONTACT_MAX_VO * 3];
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = blockIdx.x, clip = b / clip_len, bl = b - clip * clip_len;
    partials += (long)clip * HM_RED_WS_FLOATS;
    counter += (long)clip * HM_RED_WS_FLOATS;
    const float inv_cnt = 1.0f / (float)((long)clip_len * Vh);        // the mean runs over the clip's frames
    // (the picks and the hand vertices are requested before the accumulators are cleared)
    for (int i = threadIdx.x; i < 3 * Vo; i += NN_THREADS) acc[i] = 0ull;
    __syncthreads();
    float lsum = 0.f;
    for (int i = threadIdx.x; i < Vh; i += NN_THREADS) {
        const int j = nn_idx[(long)b * Vh + i];
        const float* h = vh + ((long)b * Vh + i) * 3;
        const float* o = vo + ((long)b * Vo + j) * 3;
        const float dx = o[0] - h[0], dy = o[1] - h[1], dz = o[2] - h[2];
        const float a = sqrtf(dx * dx + dy * dy + dz * dz);
        const float th = hm_tanh(a / thresh);
        lsum += thresh * th;
        const float k = (a > 0.f) ? (1.0f - th * th) / a * inv_cnt : 0.f;     // d val / d a  / a
        const float g[3] = {-k * dx, -k * dy, -k * dz};                       // d a / d h = -diff / a
        float* gh = g_hand + ((long)b * Vh + i) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gh[c] = g[c];
            atomicAdd(&acc[3 * j + c], (unsigned long long)(long long)__float2ll_rn(g[c] * CONTACT_FIX));
        }
    }
    lsum = hm_block_sum(lsum, red);          // (its barriers also order the atomics above before the reads below)
    float* go = g_obj + (long)b * Vo * 3;
    for (int i = threadIdx.x; i < 3 * Vo; i += NN_THREADS)
        go[i] = -(float)(long long)acc[i] * (1.0f / CONTACT_FIX);
    if (threadIdx.x == 0) hm_partial_store(partials + bl, lsum);
    if (hm_last_block(counter, clip_len, &s_flag)) {
        const float t = hm_last_block_sum(partials, clip_len, 1, red);
        if (threadIdx.x == 0) out[(long)clip * out_stride] = t * inv_cnt;
    }
}

extern "C" {
// workspace: reuse hm_reduce_workspace_bytes() layout (partials + counter), one slice per clip; needs
// clip frames * ceil(Vh/128) <= 512 partial floats.
int hm_nn_fwd_rigid_clips(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
                          float* metric_out, void* workspace, int clip_len, int out_stride, const int* obj_order,
                          const float* obj_spheres, const float* obj_rot6d, const float* obj_trans, const float* obj_scale,
                          const int* hand_order, int* nn_seed, hipStream_t stream);
// Scheduling hint, no effect on results: bytes of unused dynamic LDS added to the metric-only search launches.  The search is
// a chain of dependent loads in small workgroups (24 registers, 4.6 KB of LDS): eight of them fit on a CU and then hold ALL
// its wave slots while they wait - in an 8-clip batch (1680 workgroups) the line expansion of the silhouette chain, which
// runs next to it, took 250 us instead of 170.  64 KB of ballast = two search workgroups per CU.  Per calling thread (thread-local), read when
// the search is called (or captured).  Returns the previous value; bytes < 0 only queries.
static thread_local int g_nn_lds_pad = 0;
int hm_tune_nn_lds_pad(int bytes)
{
    const int prev = g_nn_lds_pad;
    if (bytes >= 0) g_nn_lds_pad = bytes;
    return prev;
}
int hm_nn_fwd_clips(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
                    float* metric_out, void* workspace, int clip_len, int out_stride, const int* obj_order,
                    hipStream_t stream)
{
    return hm_nn_fwd_rigid_clips(verts_hand, verts_obj, B, Vh, Vo, nn_idx, nn_d2, metric_out, workspace, clip_len, out_stride,
                                 obj_order, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// metric-only calls on a RIGID object: obj_spheres (B, ceil(Vo/64), 4) = centre + radius, in MESH space, of the groups of 64
// vertices taken in `obj_order`; obj_rot6d (B,3,2) / obj_trans (B,3) / obj_scale (one per clip, used as |s|) = the transform
// that produced verts_obj (hm_rigid_fwd with abs_scale); hand_order (Vh) optional: a permutation of the hand's vertices (a spatial
// sort of the template keeps the 128 vertices of a workgroup one patch of the hand); nn_seed (optional, metric-only calls,
// (2 + ceil(Vh / 128)) * B ints, zero-filled once, handed to every call of a loop): the vertex pair that held each frame's minimum
// at the previous call - its current distance bounds this call's minimum before anything is scanned (nn_min_body).  Scheduling
// data only: the result is the exact minimum.
int hm_nn_fwd_rigid_clips(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
                          float* metric_out, void* workspace, int clip_len, int out_stride, const int* obj_order,
                          const float* obj_spheres, const float* obj_rot6d, const float* obj_trans, const float* obj_scale,
                          const int* hand_order, int* nn_seed, hipStream_t stream)
{
    HM_CHECK_ARG(!obj_spheres || (obj_rot6d && obj_trans && obj_scale));
    // nn_idx == nn_d2 == NULL: metric only (exact, with pruning of the object-vertex groups that cannot hold the minimum)
    HM_CHECK_ARG(verts_hand && verts_obj && metric_out && workspace && B > 0 && Vh > 0 && Vo > 0 && (!nn_idx == !nn_d2));
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    const int nchunk = hm_cdiv(Vh, NN_HV);   // 128 hand vertices per workgroup
    const int Bc = clip_len ? clip_len : B;
    if ((long)Bc * nchunk > 512) return HM_ERR_UNSUPPORTED;
    if (!nn_idx && Vo <= 64 * NN_MAX_GROUPS)
        hipLaunchKernelGGL(k_nn_min, dim3(nchunk, B), dim3(64 * NN_WAVES), g_nn_lds_pad, stream, verts_hand, verts_obj, B, Vh, Vo,
                           (float*)workspace, (unsigned int*)((float*)workspace + 512), metric_out, Bc, out_stride, obj_order,
                           obj_spheres, obj_rot6d, obj_trans, obj_scale, hand_order, nn_seed);
    else {
        if (!nn_idx) return HM_ERR_UNSUPPORTED;     // metric-only search: <= 4096 object vertices
        hipLaunchKernelGGL(k_nn, dim3(nchunk, B), dim3(64 * NN_WAVES), 0, stream, verts_hand, verts_obj, B, Vh, Vo, nn_idx,
                           nn_d2, (float*)workspace, (unsigned int*)((float*)workspace + 512), metric_out, Bc, out_stride);
    }
    return hm_launch_status();
}
int hm_nn_fwd(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
              float* metric_out, void* workspace, hipStream_t stream)
{
    return hm_nn_fwd_clips(verts_hand, verts_obj, B, Vh, Vo, nn_idx, nn_d2, metric_out, workspace, 0, 0, nullptr, stream);
}
int hm_contact_fwd_clips(const float* verts_hand, const float* verts_obj, const int* nn_idx, int B, int Vh, int Vo,
                         float thresh, float* g_hand, float* g_obj, float* out1, void* workspace, int clip_len,
                         int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(verts_hand && verts_obj && nn_idx && g_hand && g_obj && out1 && workspace);
    HM_CHECK_ARG(B > 0 && Vh > 0 && Vo > 0 && HM_CLIP_LEN_OK(B, clip_len));
    const int Bc = clip_len ? clip_len : B;
    HM_CHECK_ARG(Bc <= 512);
    if (Vo <= CONTACT_MAX_VO) {          // one range of object vertices: both sides in one launch
        hipLaunchKernelGGL(k_contact_both, dim3(B), dim3(NN_THREADS), 0, stream, verts_hand, verts_obj, nn_idx, B, Vh, Vo,
                           thresh, g_hand, g_obj, (float*)workspace, (unsigned int*)((float*)workspace + 512), out1, Bc,
                           out_stride);
        return hm_launch_status();
    }
    hipLaunchKernelGGL(k_contact_hand, dim3(B), dim3(NN_THREADS), 0, stream, verts_hand, verts_obj, nn_idx, B, Vh, Vo,
                       thresh, g_hand, (float*)workspace, (unsigned int*)((float*)workspace + 512), out1, Bc, out_stride);
    hipLaunchKernelGGL(k_contact_obj, dim3(B, hm_cdiv(Vo, CONTACT_MAX_VO)), dim3(NN_THREADS), 0, stream, nn_idx, g_hand, B, Vh,
                       Vo, g_obj);
    return hm_launch_status();
}
int hm_contact_fwd(const float* verts_hand, const float* verts_obj, const int* nn_idx, int B, int Vh, int Vo,
                   float thresh, float* g_hand, float* g_obj, float* out1, void* workspace, hipStream_t stream)
{
    return hm_contact_fwd_clips(verts_hand, verts_obj, nn_idx, B, Vh, Vo, thresh, g_hand, g_obj, out1, workspace, 0, 0,
                                stream);
}
#ifdef NN_PHASES
int hm_debug_nn_phases(unsigned long long* out)          // 10 counters, then cleared (debug build -DNN_PHASES)
{
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_ph), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_ph), z, sizeof(z));
    return HM_OK;
}
#endif
}  // extern "C"
