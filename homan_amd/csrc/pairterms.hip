// pairterms.hip -- the small losses that need BOTH vertex sets of a frame (or only wait for the object's), in ONE launch.
//
// At one clip the iteration is a chain of short, latency-bound launches on each of its two streams, and every launch costs
// its own dependent memory round trips plus a graph edge (~5-10 us).  The object's temporal smoothness (reference
// homan/lossutils.py:18-36), the coarse interaction term (homan/losses.py:199-242) and the metric-only nearest-vertex search
// (homan/losses.py:225-241) all start from the same two buffers and feed nothing to each other: here they are block ranges
// of one grid - [search | interaction | smoothness] - running side by side, each with its own reduce workspace.  The bodies
// are the ones of k_nn_min / k_inter / k_smooth (pair_bodies.h): same floats as the three separate launches.
#include "hm_common.h"
#include "pair_bodies.h"

struct HandTerms {          // arguments of hm_hand_terms_fwd_clips (verts = the launch's hand vertices); out_v2d == NULL: none
    const float* ref2d; float image_size; float* unit_v2d; float* out_v2d; float* unit_smooth; float* out_smooth;
    const float* pca; long npca; const float* s_obj; const float* m_obj; const float* s_hand; const float* m_hand;
    float* g_pca; float* g_sobj; float* g_shand; float* out_priors; float* partials; unsigned int* counter; int nblk;
};
__global__ __launch_bounds__(256) void k_pair_terms(
    const float* __restrict__ vh, const float* __restrict__ vo, const float* __restrict__ camintr, int B, int Vh, int Vo,
    int clip_len, int out_stride,
    // search (n_nn blocks = nchunk x B; metric_out == NULL: none)
    int nchunk, float* __restrict__ nn_blockmin, unsigned int* nn_counter, float* __restrict__ metric_out,
    const int* __restrict__ obj_order,
    // interaction (B blocks; out_inter == NULL: none)
    float expansion, float zthresh, float* __restrict__ frame_rec, unsigned int* inter_counter, float* __restrict__ out_inter,
    // smoothness of the object vertices (sm_nblk x clips blocks; out_smooth == NULL: none)
    int sm_nblk, float* __restrict__ unit_smooth, float* __restrict__ sm_partials, unsigned int* sm_counter,
    float* __restrict__ out_smooth, HandTerms ht, int clips, const float* __restrict__ sph_mesh,
    const float* __restrict__ obj_rot6d, const float* __restrict__ obj_trans, const float* __restrict__ obj_scale,
    const int* __restrict__ hand_order, int* __restrict__ nn_idx, float* __restrict__ nn_d2, int* __restrict__ nn_seed)
{
    HM_HAND_KERNEL();
    int i = blockIdx.x;
    const int n_nn = metric_out ? nchunk * B : 0, n_in = out_inter ? B : 0;
    if (i < n_nn) {
        if (nn_idx)       // the full search (nearest object vertex of every hand vertex, for the contact term) instead
            nn_full_body(vh, vo, B, Vh, Vo, nn_idx, nn_d2, nn_blockmin, nn_counter, metric_out, clip_len, out_stride, i % nchunk,
                         i / nchunk, nchunk);
        else
            nn_min_body(vh, vo, B, Vh, Vo, nn_blockmin, nn_counter, metric_out, clip_len, out_stride, obj_order, i % nchunk,
                        i / nchunk, nchunk, sph_mesh, obj_rot6d, obj_trans, obj_scale, hand_order, nn_seed);
        return;
    }
    i -= n_nn;
    if (i < n_in) {
        inter_body(vh, vo, camintr, B, Vh, Vo, expansion, zthresh, frame_rec, inter_counter, out_inter, clip_len, out_stride, i);
        return;
    }
    i -= n_in;
    const int n_ht = ht.out_v2d ? ht.nblk * clips : 0;
    if (i < n_ht) {
        hand_terms_body(vh, camintr, 1, ht.ref2d, ht.image_size, clip_len, Vh, ht.unit_v2d, ht.out_v2d, ht.unit_smooth,
                        ht.out_smooth, ht.pca, ht.npca, ht.s_obj, ht.m_obj, ht.s_hand, ht.m_hand, ht.g_pca, ht.g_sobj, ht.g_shand,
                        ht.out_priors, ht.partials, ht.counter, out_stride, i % ht.nblk, i / ht.nblk, ht.nblk);
        return;
    }
    i -= n_ht;
    smooth_body(vo, clip_len, Vo, 1, unit_smooth, sm_partials, sm_counter, out_smooth, out_stride, i % sm_nblk, i / sm_nblk,
                sm_nblk);
}

extern "C" {
// One launch for up to three terms of the frames' (hand, object) vertex pairs; every term is optional (its output pointer
// NULL) and equals its own entry point on the same inputs:
//   metric_out   -> hm_nn_fwd_clips(..., nn_idx = nn_d2 = NULL, ...)   [<= 4096 object vertices]   workspace ws_nn
//                   (nn_idx / nn_d2 != NULL: hm_nn_fwd_clips WITH them - the full search the contact term needs)
//   out_inter    -> hm_inter_fwd_clips(...)                             frame records `frame_rec`    workspace ws_inter
//   out_smooth   -> hm_smooth_fwd_clips(verts_obj, ..., hand_nb = 1)    unit gradient `unit_smooth`  workspace ws_smooth
//   ht_out_v2d2  -> hm_hand_terms_fwd_clips(verts_hand, camintr, hand_nb = 1, ...) (one hand per frame)      workspace ws_hand
//   obj_spheres / obj_rot6d / obj_trans / obj_scale / hand_order: optional, see hm_nn_fwd_rigid_clips (scheduling data of the search)
// The reduce workspaces (hm_reduce_workspace_bytes() per clip each) must be distinct: the terms run concurrently.
int hm_pair_terms_fwd_clips(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                            float* metric_out, const int* obj_order, void* ws_nn, float expansion, float zthresh,
                            float* frame_rec, float* out_inter, void* ws_inter, float* unit_smooth, float* out_smooth,
                            void* ws_smooth,
                            const float* ht_ref2d, float ht_image_size, float* ht_unit_v2d, float* ht_out_v2d2,
                            float* ht_unit_smooth, float* ht_out_smooth1, const float* ht_pca, long ht_npca,
                            const float* ht_s_obj, const float* ht_m_obj, const float* ht_s_hand, const float* ht_m_hand,
                            float* ht_g_pca, float* ht_g_sobj, float* ht_g_shand, float* ht_out_priors3, void* ws_hand,
                            const float* obj_spheres, const float* obj_rot6d, const float* obj_trans, const float* obj_scale,
                            const int* hand_order, int* nn_idx, float* nn_d2, int* nn_seed, int clip_len, int out_stride,
                            hipStream_t stream)
{
    HM_CHECK_ARG((!nn_idx == !nn_d2) && (!nn_idx || metric_out));
    HM_CHECK_ARG(!obj_spheres || (obj_rot6d && obj_trans && obj_scale));
    HM_CHECK_ARG(verts_hand && verts_obj && B > 0 && Vh > 0 && Vo > 0 && HM_CLIP_LEN_OK(B, clip_len));
    HM_CHECK_ARG(metric_out || out_inter || out_smooth || ht_out_v2d2);
    HM_CHECK_ARG(!ht_out_v2d2 || (ht_ref2d && ht_unit_v2d && ht_unit_smooth && ht_out_smooth1 && ws_hand && camintr));
    HM_CHECK_ARG(!ht_pca || (ht_npca > 0 && ht_s_obj && ht_m_obj && ht_s_hand && ht_m_hand && ht_g_pca && ht_g_sobj &&
                             ht_g_shand && ht_out_priors3));
    HM_CHECK_ARG((!metric_out || ws_nn) && (!out_inter || (ws_inter && frame_rec && camintr)) &&
                 (!out_smooth || (ws_smooth && unit_smooth)));
    const int Bc = clip_len ? clip_len : B, clips = B / Bc;
    const int nchunk = hm_cdiv(Vh, NN_HV);
    if (metric_out && ((long)Bc * nchunk > 512 || (!nn_idx && Vo > 64 * NN_MAX_GROUPS))) return HM_ERR_UNSUPPORTED;
    if (out_inter && Bc > 512) return HM_ERR_UNSUPPORTED;
    const int sm_nblk = min(256, hm_cdiv((long)Bc * Vo * 3, RED_THREADS * 4));       // = hm_smooth_fwd_clips' grid
    HandTerms ht = {ht_ref2d, ht_image_size, ht_unit_v2d, ht_out_v2d2, ht_unit_smooth, ht_out_smooth1, ht_pca, ht_npca,
                    ht_s_obj, ht_m_obj, ht_s_hand, ht_m_hand, ht_g_pca, ht_g_sobj, ht_g_shand, ht_out_priors3, (float*)ws_hand,
                    ws_hand ? (unsigned int*)((float*)ws_hand + 512) : nullptr,
                    min(170, hm_cdiv((long)Bc * Vh * 3, RED_THREADS * 2))};       // = hm_hand_terms_fwd_clips' grid
    const int blocks = (metric_out ? nchunk * B : 0) + (out_inter ? B : 0) + (ht_out_v2d2 ? ht.nblk * clips : 0) +
                       (out_smooth ? sm_nblk * clips : 0);
    hipLaunchKernelGGL(k_pair_terms, dim3(blocks), dim3(256), g_hm_lds_pad[HM_PAD_PAIR_TERMS], stream, verts_hand, verts_obj, camintr, B, Vh, Vo, Bc,
                       out_stride, nchunk, (float*)ws_nn, ws_nn ? (unsigned int*)((float*)ws_nn + 512) : nullptr, metric_out,
                       obj_order, expansion, zthresh, frame_rec,
                       ws_inter ? (unsigned int*)((float*)ws_inter + 512) : nullptr, out_inter, sm_nblk, unit_smooth,
                       (float*)ws_smooth, ws_smooth ? (unsigned int*)((float*)ws_smooth + 512) : nullptr, out_smooth, ht, clips,
                       obj_spheres, obj_rot6d, obj_trans, obj_scale, hand_order, nn_idx, nn_d2, nn_seed);
    return hm_launch_status();
}
}  // extern "C"
