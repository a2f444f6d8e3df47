// raster_api.hip -- C ABI of the NMR-semantics silhouette rasteriser for CDNA4 (gfx950), forward + pseudo-gradient.
// The kernels live in raster_setup / raster_fwd / raster_lines / raster_sweep / raster_depth .hip over raster_common.h (device
// helpers, records) and raster_ws.h (workspace layout, launchers); this unit holds the entry points of include/homan_amd.h.
//
// Replaces, on the reference's hot path, the third-party CUDA extension `neural_renderer`
// as called from reference homan/losses.py:187 (Renderer(...)(verts, faces, K=, mode="silhouettes"))
// with the ctor defaults of homan/losses.py:73-77 (anti_aliasing, fill_back, near=0.1, far=100,
// eps=1e-3), and fuses the masked-MSE / IoU reduction of homan/losses.py:188-197.
//
// Semantics (bit-compatible with oracle/csrc/nmr_raster.c given identical NDC faces):
//   hard z-buffer coverage on a (2S)^2 sample grid, inclusive edge test, perspective-correct
//   z from clamped+renormalised barycentrics, strict z-min with lowest-face-index tie-break,
//   fill_back (both windings, index f and F+f), vertical flip, 2x2 average pool; backward =
//   per-(face,edge,axis) line sweeps comparing in/out alpha (Kato et al. 2018).
//
// Design (CDNA4), see DESIGN.md section 4 for the measurements behind each choice:
//   forward   k_setup_faces (thread / face: optional rigid transform, projection, tight sample box, super-region bins; extra
//             workgroups write the camera-space vertices) -> k_raster_fwd (workgroup = 32x32-sample region: candidates of
//             its bin split into the camera-facing winding class and the hidden one; (candidate, 4x4 block) units
//             flattened over the threads; visibility by ds_min_u64 on an LDS z-buffer = the strict z test in ascending face
//             order, bit-exact; hidden-class units filtered against per-block depths and run on full waves of survivors;
//             epilogue per 8x8 output tile: index map, pooled silhouette, fused masked-MSE / IoU terms, sweep bit planes);
//   backward  k_bwd_lines (bit lines -> position-sorted source arrays + per-word records + per-line summaries; its first
//             workgroups build the flattened work list of the sweeps) -> k_bwd_sweep (persistent waves over 256-item
//             units: summary filter, then owner tests / slices / (item, source) pairs on full waves) -> per-corner NDC
//             gradients, gathered per vertex by k_bwd_gather or inside hm_rigid_bwd_sil.
#include "raster_ws.h"

RasterTune& hm_raster_tune()
{
    static thread_local RasterTune t = {0, 0, SWEEP_BLOCKS};
    return t;
}
int& hm_sweep_cap_override()
{
    static int cap = 0;
    return cap;
}
// measurement hook (hm_debug_sil_timing): HIP events recorded on the launch stream right before / after the three heavy
// kernels of the silhouette chain, so a caller that drives the optimisation loop launch by launch (not from a captured
// graph) reads the duration each kernel had INSIDE the loop, next to whatever runs on the other streams
static hipEvent_t g_tev[6];
static int g_timing = 0;
#define HM_TIME_MARK(k, stream) do { if (g_timing) (void)hipEventRecord(g_tev[k], stream); } while (0)

extern "C" {

size_t hm_sil_workspace_bytes(int B, int V, int F, int S)
{
    const size_t is = 2 * (size_t)S;
    size_t n = 256;                             // counter word (zero-initialised by the caller once)
    n += al256((size_t)B * 16);                 // per-frame reduction records
    n += al256((size_t)B * V * 3 * 4);          // ndc
    n += al256((size_t)B * F * 9 * 4);          // faces9
    n += al256((size_t)B * F * 8);              // boxes
    n += al256((size_t)B * is * is * 4);        // idx_map
    n += al256((size_t)B * is * (is / 16) * 2); // alpha16
    n += al256((size_t)B * S * S * 4);          // dimg
    n += al256((size_t)B * (S / 8) * (S / 8) * 16); // partials
    n += al256((size_t)B * is * is * 4);        // gimg (pooled grid, or the full sample grid without anti-aliasing)
    n += al256((size_t)B * is * (is / 16) * 4); // row masks, 2 planes
    n += al256((size_t)B * is * (is / 16) * 4); // column masks, 2 planes
    n += al256((size_t)B * F * 24 * 4);         // parts
    n += al256((size_t)B * F * 2);              // owned
    n += al256((size_t)B * SR_MAX * 8);                        // super-region bin counters + their ticket words
    n += al256((size_t)B * (S / 16) * (S / 16));               // per-region "outputs hold the empty pattern" flags
    n += al256((size_t)B * SR_MAX * F * 4);                    // super-region face lists (worst case: every face in every bin)
    n += al256(4 * (size_t)B * is * (is / 64) * 16);            // per-line records {mask word, sources before it}
    n += al256((size_t)B * 2 * is * 16);                        // per-line summaries {first, last+1, word mask} x 2 planes
    n += al256(4 * (size_t)B * is * is * sizeof(SweepSrc));     // per-line source arrays (2 planes x 2 orientations)
    n += al256((size_t)B * F * sizeof(SweepFace));              // sweep work list: face records,
    n += al256((size_t)B * F * 4) * 2;                          //   their first items, their tickets,
    n += al256(sweep_ucap(B, F) * 4);                           //   first face of every unit,
    n += al256(sweep_slot_cap(B, F) * 24);                      //   per-unit partials of faces spread over several units
    n += al256(ts_units(B, F, S) * 16);                          // in-graph timestamps (hm_sil_timestamps)
    n += al256(ts_raster_units(B, S) * 4) * 3;                   // adaptive raster launch order: entries, scratch, times
    return n;
}

// Scheduling hint, no effect on results: bytes of unused dynamic LDS added to every k_raster_fwd launch.  The rasteriser's
// 24.9 KB of LDS and 80 registers fill a CU with 6 workgroups and leave nothing for the kernels of the caller's other
// stream (72 registers for the MANO forward: it then waits for the raster's tail); 3 KB of ballast caps the CU at 5
// workgroups.  Per calling thread (thread-local); read when hm_sil_fwd is called (or captured).  Returns the previous value; < 0 only queries.
int hm_tune_raster_lds_pad(int bytes)
{
    const int prev = hm_raster_tune().raster_lds_pad;
    if (bytes >= 0) hm_raster_tune().raster_lds_pad = bytes;
    return prev;
}
// Scheduling hint, no effect on results: adaptive launch order of the forward raster (RasterTune::raster_reorder).  enable > 0: on,
// 0: off, < 0: query.  Returns the previous value.  Per calling thread (thread-local); read when hm_sil_fwd / hm_sil_bwd are called (or captured).
int hm_tune_raster_reorder(int enable)
{
    const int prev = hm_raster_tune().raster_reorder;
    if (enable >= 0) hm_raster_tune().raster_reorder = enable ? 1 : 0;
    return prev;
}

// Scheduling hint, no effect on results: number of persistent workgroups of the edge-sweep kernel (default 1280 = 5 per
// CU).  The sweeps share the GPU with whatever runs on the caller's other streams; a loop whose other stream is the
// longer chain (collision + contact terms) finishes sooner with fewer sweep workgroups (768).  Per calling thread (thread-local); read when
// hm_sil_bwd is called (or captured).  Returns the previous value; blocks <= 0 only queries.
int hm_tune_sweep_blocks(int blocks)
{
    const int prev = hm_raster_tune().sweep_blocks;
    if (blocks > 0) hm_raster_tune().sweep_blocks = blocks;
    return prev;
}

// Forward: silhouettes (B,S,S) of `verts` under per-frame intrinsics K, optional fused masked-MSE/IoU.
//   keep/ref/keep_sum/loss_out may be NULL (render only).  loss_out[0]=loss_sil, loss_out[1]=mean IoU.
//   *_clips: the B frames are B / clip_len clips of clip_len frames (0: one clip); keep_sum and rigid_scale hold one
//   entry per clip, the loss / IoU of clip c go to loss_out[c * out_stride + 0 / 1].
//   *_phase_clips: the same forward in two calls for a caller that forks a second stream off the camera-space vertices:
//   phases = 1 launches the face setup only (cam_verts_out is complete when it ends), 2 the rasteriser (+ reduction) only,
//   3 both (= hm_sil_fwd_clips); both calls take the same arguments.
int hm_sil_fwd_phase_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                           float orig_size, float znear, float zfar, const float* keep, const float* ref,
                           const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                           float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                           const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                           int out_stride, float* cam_verts_out, int phases, hipStream_t stream)
{
    HM_CHECK_ARG(phases >= 1 && phases <= 3);
    HM_CHECK_ARG(verts && faces && K && pooled && workspace && HM_CLIP_LEN_OK(B, clip_len));
    if (clip_len == 0) clip_len = B;
    HM_CHECK_ARG(!rigid_rot6d || (rigid_trans && rigid_scale));
    HM_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    if (S % 16 != 0 || 2 * S > 8192 || 2L * F >= (1L << 30) || B >= 32768) return HM_ERR_UNSUPPORTED;
    HM_CHECK_ARG(faces_bstride == 0 || faces_bstride == 3 * F);
    SilWs w = carve(workspace, B, V, F, S);
    const int is = 2 * S;
    int* bins = is <= (SR_MAX == 64 ? 1024 : 0) ? w.bin_cnt : nullptr;      // <= SR_MAX super-regions per frame
    HM_CHECK_ARG(!cam_verts_out || rigid_rot6d);
    if (phases & 1)
        hm_launch_setup_faces(w, verts, K, orig_size, faces, faces_bstride, B, V, F, is, bins, rigid_rot6d, rigid_trans, rigid_scale,
                              rigid_abs, clip_len, cam_verts_out, stream);
    if (!(phases & 2)) return hm_launch_status();
    const bool fused = keep && ref;
    const RasterTune& tune = hm_raster_tune();
    RasterFwdArgs a = {B, F, S, znear, zfar, pooled, keep, ref, fused, work_order, pooled_depth, bins, 1, persistent_outputs,
                       alpha_full, mask_shared, fused && (alpha_full || (mask_shared & 2)), tune.raster_reorder != 0,
                       tune.raster_lds_pad};
    HM_TIME_MARK(0, stream);
    hm_launch_raster_fwd(w, a, stream);
    HM_TIME_MARK(1, stream);
    if (fused && keep_sum && loss_out) hm_launch_sil_reduce(w, B, S, keep_sum, loss_out, nullptr, clip_len, out_stride, stream);
    return hm_launch_status();
}
int hm_sil_fwd_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                     float orig_size, float znear, float zfar, const float* keep, const float* ref,
                     const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                     float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                     const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                     int out_stride, float* cam_verts_out, hipStream_t stream)
{
    return hm_sil_fwd_phase_clips(verts, faces, faces_bstride, K, B, V, F, S, orig_size, znear, zfar, keep, ref, keep_sum, pooled,
                                  loss_out, work_order, pooled_depth, alpha_full, mask_shared, rigid_rot6d, rigid_trans,
                                  rigid_scale, rigid_abs, persistent_outputs, workspace, clip_len, out_stride, cam_verts_out, 3,
                                  stream);
}
int hm_sil_fwd(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
               float orig_size, float znear, float zfar, const float* keep, const float* ref,
               const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
               float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
               const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, hipStream_t stream)
{
    return hm_sil_fwd_clips(verts, faces, faces_bstride, K, B, V, F, S, orig_size, znear, zfar, keep, ref, keep_sum, pooled,
                            loss_out, work_order, pooled_depth, alpha_full, mask_shared, rigid_rot6d, rigid_trans,
                            rigid_scale, rigid_abs, persistent_outputs, workspace, 0, 0, nullptr, stream);
}

// Several renders - each with its own mesh, cameras, outputs and workspace - as ONE face-setup launch and ONE raster launch
// (k_setup_faces_multi / k_raster_fwd_multi: the kernels' bodies, per render the arguments hm_sil_fwd_clips would pass).  What a
// caller gains is launches: the depth term of reference homan/homan.py:384-419 renders the object twice per iteration (ROI
// silhouette, full-image depth) and the hand once, and clips of different meshes (reference homan/datasets/core50.py:22-42)
// are renders of different (V, F) - per-render vertex / face arrays ARE the per-frame mesh offsets.  Results: those of n
// separate hm_sil_fwd_clips calls, bit for bit (every workgroup runs the same code on the same data); each render's backward
// (hm_sil_bwd_clips / hm_depth_bwd on ITS workspace) is unchanged.  phases as hm_sil_fwd_phase_clips.
size_t hm_sil_render_bytes(void) { return sizeof(HmSilRender); }
int hm_sil_fwd_multi(const HmSilRender* renders, int n, int phases, hipStream_t stream)
{
    HM_CHECK_ARG(renders && n >= 1 && n <= HM_MAX_RENDERS && phases >= 1 && phases <= 3);
    SilWs ws[HM_MAX_RENDERS];
    SetupFacesArgs sa[HM_MAX_RENDERS];
    RasterFwdArgs ra[HM_MAX_RENDERS];
    const RasterTune& tune = hm_raster_tune();
    for (int g = 0; g < n; ++g) {
        const HmSilRender& r = renders[g];
        HM_CHECK_ARG(r.verts && r.faces && r.K && r.pooled && r.workspace && HM_CLIP_LEN_OK(r.B, r.clip_len));
        HM_CHECK_ARG(!r.rigid_rot6d || (r.rigid_trans && r.rigid_scale));
        HM_CHECK_ARG(r.B > 0 && r.V > 0 && r.F > 0 && r.S > 0 && (!r.keep == !r.ref));
        if (r.S % 16 != 0 || 2 * r.S > 8192 || 2L * r.F >= (1L << 30) || r.B >= 32768) return HM_ERR_UNSUPPORTED;
        HM_CHECK_ARG(r.faces_bstride == 0 || r.faces_bstride == 3 * r.F);
        HM_CHECK_ARG(!r.cam_verts_out || r.rigid_rot6d);
        for (int q = 0; q < g; ++q) HM_CHECK_ARG(renders[q].workspace != r.workspace);      // a workspace carries ONE render
        ws[g] = carve(r.workspace, r.B, r.V, r.F, r.S);
        const int is = 2 * r.S;
        int* bins = is <= (SR_MAX == 64 ? 1024 : 0) ? ws[g].bin_cnt : nullptr;
        const SetupFacesArgs s1 = {r.verts, r.K, r.orig_size, r.faces, r.faces_bstride, r.B, r.V, r.F, is, bins, r.rigid_rot6d,
                                   r.rigid_trans, r.rigid_scale, r.rigid_abs, r.clip_len ? r.clip_len : r.B, r.cam_verts_out};
        sa[g] = s1;
        const bool fused = r.keep && r.ref;
        const RasterFwdArgs r1 = {r.B, r.F, r.S, r.znear, r.zfar, r.pooled, r.keep, r.ref, fused, r.work_order, r.pooled_depth, bins, 1,
                                  r.persistent_outputs, nullptr, r.mask_shared & 1, false, tune.raster_reorder != 0, tune.raster_lds_pad};
        ra[g] = r1;
    }
    if (phases & 1) hm_launch_setup_faces_multi(ws, sa, n, stream);
    if (phases & 2) hm_launch_raster_fwd_multi(ws, ra, n, stream);
    return hm_launch_status();
}

// Scheduling hint, no effect on results: which winding class of the mesh (0: faces as stored, 1: reversed copies of
// fill_back) holds the camera-facing surface.  The forward rasterises that class first and tests the units of the other
// class against per-block hidden depths before doing any per-sample work: on a closed mesh the far class owns nothing.
// Stored in the workspace (stream-ordered 4-byte write); the zero-filled default is class 0.
int hm_sil_hint_near_winding(void* workspace, int winding, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && (winding == 0 || winding == 1));
    return hipMemsetAsync((char*)workspace + 24 * 4, winding, 1, stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}

// The loss / IoU reduction of a forward that was called with keep/ref but loss_out == NULL: the backward does not
// depend on it, so a caller with a second stream takes it off the critical path.
// frame_out (B,2) optional: per-frame {sum of squares (un-normalised), IoU}; loss_out may then be NULL.
int hm_sil_reduce_clips(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out,
                        void* workspace, int clip_len, int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && S > 0 && (frame_out || loss_out) && (!loss_out || keep_sum));
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    SilWs w = carve(workspace, B, V, F, S);
    hm_launch_sil_reduce(w, B, S, keep_sum, loss_out, frame_out, clip_len ? clip_len : B, out_stride, stream);
    return hm_launch_status();
}
int hm_sil_reduce(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out, void* workspace,
                  hipStream_t stream)
{
    return hm_sil_reduce_clips(B, V, F, S, keep_sum, loss_out, frame_out, workspace, 0, 0, stream);
}

// Backward.  mode 1 (fused loss): upstream = d/d loss_sil (device scalar), uses dimg from the forward.
//            mode 2: as mode 1, and the caller guarantees upstream[0] > 0 (one launch less).
//            mode 0 (render):     grad_pooled (B,S,S) = dL/d silhouettes.
//            mode 3 (render without anti-aliasing): grad_pooled is (B,2S,2S) = dL/d alpha_full.
//            mode 4 (fused per-sample L2 of a forward called with alpha_full + keep/ref): upstream (B) = dL/d frame sums, all > 0.
// adjacency (CSR over V) describes the shared face topology.  grad_verts (B,V,3) is overwritten.
//   *_clips (modes 1 / 2): keep_sum holds one entry per clip of clip_len frames and the 1/B of the loss is 1/clip_len;
//   `upstream` stays one scalar shared by the clips.  loss_out (optional, modes 1 / 2): the loss / IoU reduction of a forward
//   called with keep / ref but loss_out == NULL (what hm_sil_reduce_clips computes, same arithmetic) rides at the front of
//   the backward's first launch: clip c's values at loss_out[c * out_stride + 0 / 1].
// phases: bit 0 = sample-gradient masks (generic modes) + line expansion + work list, bit 1 = edge sweeps + vertex gather: the
// backward in two calls for a caller that lets other streams wait for the END of the line expansion (the kernel of the chain
// that suffers most from latency-bound neighbours holding its wave slots); hm_sil_bwd_clips = both
int hm_sil_bwd_phase_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                           const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                           const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                           int clip_len, float* loss_out, int out_stride, int phases, int sum_log2q, hipStream_t stream)
{
    HM_CHECK_ARG(!loss_out || ((mode == 1 || mode == 2) && keep_sum));
    HM_CHECK_ARG(sum_log2q <= 0 && sum_log2q >= -60);
    HM_CHECK_ARG(verts && K && adj_off && adj_items && workspace);        // grad_verts == NULL: no vertex gather (see hm_sil_parts)
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    if (clip_len == 0) clip_len = B;
    HM_CHECK_ARG((mode == 0 || mode == 3) ? grad_pooled != nullptr : ((mode == 4 || mode == 5) ? upstream != nullptr : (upstream && keep_sum)));
    HM_CHECK_ARG(mode >= 0 && mode <= 5);
    if (S % 32 != 0 || S > 32 * SWEEP_CUMW) return HM_ERR_UNSUPPORTED;     // 64-sample mask words, <= SWEEP_CUMW per line
    SilWs w = carve(workspace, B, V, F, S);
    HM_CHECK_ARG(phases >= 1 && phases <= 3);
    if ((phases & 1) && mode != 2 && mode != 4 && mode != 5)      // modes 2 / 4 / 5: the caller guarantees upstream > 0, the forward's planes are the backward's
        hm_launch_bwd_masks(w, mode == 1 ? w.dimg : grad_pooled, mode, upstream, keep_sum, B, S, clip_len, stream);
    HM_TIME_MARK(2, stream);
    if (phases & 1) hm_launch_lines(w, B, F, S, mode, upstream, keep_sum, clip_len, stream, loss_out, out_stride);
    HM_TIME_MARK(3, stream);
    if (!(phases & 2)) return hm_launch_status();
    hm_launch_sweep(w, B, F, S, eps, sum_log2q, stream);
    HM_TIME_MARK(4, stream);
    if (grad_verts) hm_launch_bwd_gather(w, adj_off, adj_items, verts, K, B, V, F, orig_size, grad_ndc, grad_verts, stream);
    return hm_launch_status();
}
int hm_sil_bwd_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                     const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                     const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                     int clip_len, float* loss_out, int out_stride, int sum_log2q, hipStream_t stream)
{
    return hm_sil_bwd_phase_clips(verts, K, B, V, F, S, orig_size, eps, mode, upstream, grad_pooled, keep_sum, adj_off, adj_items,
                                  face_order, grad_verts, grad_ndc, workspace, clip_len, loss_out, out_stride, 3, sum_log2q, stream);
}
int hm_sil_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
               const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
               const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
               int sum_log2q, hipStream_t stream)
{
    return hm_sil_bwd_clips(verts, K, B, V, F, S, orig_size, eps, mode, upstream, grad_pooled, keep_sum, adj_off, adj_items,
                            face_order, grad_verts, grad_ndc, workspace, 0, nullptr, 0, sum_log2q, stream);
}

// (B,F,3,2) DOUBLES: d loss / d NDC (x, y) per face corner, as left by the last hm_sil_bwd (exact sums of terms on the
// grid 2^sum_log2q, see hm_quant): input of hm_rigid_bwd_sil.
const double* hm_sil_parts(const void* workspace, int B, int V, int F, int S)
{
    return carve((void*)workspace, B, V, F, S).parts;
}

// rgb image (B,3,S,S) of the last hm_sil_fwd on this workspace (same verts / faces): per-face colours `textures`
// (B,F,3) under flat lighting, as nr.renderer.Renderer.render returns it for texture_size 1 (reference
// homan/homan.py:535-538 with the light of :173-176).  light_dir / background: HOST float[3].
int hm_shade_rgb(const float* verts, const int* faces, int faces_bstride, const float* textures, int B, int V, int F, int S,
                 const float* light_dir, float intensity_ambient, float intensity_directional, const float* background,
                 float* rgb, void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts && faces && textures && light_dir && background && rgb && workspace);
    HM_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    hm_launch_shade_rgb(w, verts, faces, faces_bstride, textures, B, V, F, S, light_dir, intensity_ambient, intensity_directional,
                        background, rgb, stream);
    return hm_launch_status();
}


// Measurement hook for bench.py: runs one full forward + backward (fused-loss mode, upstream = 1) to populate the
// workspace, then `reps` launches of k_raster_fwd alone and `reps` launches of k_bwd_sweep alone, each bracketed by two
// HIP events recorded on `stream`; avg_ms[0] / avg_ms[1] (HOST pointer) receive the average launch durations in
// milliseconds.  Synchronises.
int hm_bench_sil_kernels(const float* verts, const int* faces, const float* K, int B, int V, int F, int S,
                         const float* keep, const float* ref, const float* keep_sum, float* pooled, float* loss_out,
                         const int* work_order, const int* adj_off, const int* adj_items, const int* face_order,
                         const float* upstream, float* grad_verts, void* workspace, int reps, float* avg_ms,
                         hipStream_t stream)
{
    HM_CHECK_ARG(verts && faces && K && keep && ref && keep_sum && pooled && loss_out && workspace && reps > 0 && avg_ms);
    HM_CHECK_ARG(adj_off && adj_items && upstream && grad_verts);
    int rc = hm_sil_fwd(verts, faces, 0, K, B, V, F, S, 1.0f, 0.1f, 100.0f, keep, ref, keep_sum, pooled, loss_out,
                        work_order, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, workspace, stream);
    if (rc != HM_OK) return rc;
    rc = hm_sil_bwd(verts, K, B, V, F, S, 1.0f, 1e-3f, 1, upstream, nullptr, keep_sum, adj_off, adj_items, face_order,
                    grad_verts, nullptr, workspace, 0, stream);
    if (rc != HM_OK) return rc;
    SilWs w = carve(workspace, B, V, F, S);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return HM_ERR_LAUNCH;
    float ms = 0.f;
    // the forward's last workgroup emptied the super-region bins: fill them again and keep them across the timed launches
    int* bins = 2 * S <= 1024 ? w.bin_cnt : nullptr;
    hm_launch_setup_faces(w, verts, K, 1.0f, faces, 0, B, V, F, 2 * S, bins, nullptr, nullptr, nullptr, 0, B, nullptr, stream);
    // (steady state of a fixed loop: background regions skipped; the bins are not reset between the launches)
    RasterFwdArgs a = {B, F, S, 0.1f, 100.0f, pooled, keep, ref, true, work_order, nullptr, bins, 0, 1, nullptr, 0, false, false, 0};
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) hm_launch_raster_fwd(w, a, stream);
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[0] = ms / (float)reps;
    (void)hipMemsetAsync(w.bin_cnt, 0, (size_t)B * SR_MAX * 4, stream);
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) hm_launch_sweep(w, B, F, S, 1e-3f, 0, stream);
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[1] = ms / (float)reps;
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) hm_launch_lines(w, B, F, S, 1, upstream, keep_sum, B, stream);
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[2] = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return hm_launch_status();
}

// enable != 0: from now on hm_sil_fwd / hm_sil_bwd record HIP events around k_raster_fwd, k_bwd_lines and k_bwd_sweep on
// their launch stream (do not enable while a stream capture is in progress); 0: stop and release the events.
int hm_debug_sil_timing(int enable)
{
    if (enable && !g_timing) {
        for (int k = 0; k < 5; ++k)
            if (hipEventCreate(&g_tev[k]) != hipSuccess) return HM_ERR_LAUNCH;
        g_timing = 1;
    } else if (!enable && g_timing) {
        g_timing = 0;
        for (int k = 0; k < 5; ++k) (void)hipEventDestroy(g_tev[k]);
    }
    return HM_OK;
}
// In-graph timing of the same three kernels: device wall-clock stamps stored by the workgroups themselves (see
// hm_ts_enabled above), so the numbers come from launches replayed from a captured hipGraph - where ROCm allows no events.
//   hm_sil_timestamps(ws, B, V, F, S, 1, stream): switch on and arm (two async memsets, no host synchronisation); call
//   before every replay;  (..., 0, stream): switch off
//   hm_sil_timestamps_bytes(B, V, F, S): size of one record of raw stamps
//   hm_sil_timestamps_save(ws, B, V, F, S, dst, stream): async device copy of the raw stamps to `dst` - a caller that must
//   not synchronise between replays saves every iteration's record and reads them all at the end
//   hm_sil_timestamps_read(ws, B, V, F, S, saved, us3, stream): waits for the stream; durations (earliest start to latest
//   end over the workgroups) of k_raster_fwd, k_bwd_lines, k_bwd_sweep in microseconds -> us3 (HOST), from `saved` (a
//   hm_sil_timestamps_save record) or, saved == NULL, from the workspace itself (0 for a kernel that did not run)
size_t hm_sil_timestamps_bytes(int B, int V, int F, int S) { (void)V; return ts_units(B, F, S) * 16; }
int hm_sil_timestamps(void* workspace, int B, int V, int F, int S, int enable, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && F > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    if (hipMemsetAsync((char*)workspace + 24 * 4 + 1, enable ? 1 : 0, 1, stream) != hipSuccess) return HM_ERR_LAUNCH;
    if (enable && hipMemsetAsync(w.ts, 0, ts_units(B, F, S) * 16, stream) != hipSuccess) return HM_ERR_LAUNCH;
    return HM_OK;
}
int hm_sil_timestamps_save(const void* workspace, int B, int V, int F, int S, void* dst, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && dst);
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(dst, w.ts, ts_units(B, F, S) * 16, hipMemcpyDeviceToDevice, stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}
int hm_sil_timestamps_read(const void* workspace, int B, int V, int F, int S, const void* saved, float* us3, hipStream_t stream)
{
    HM_CHECK_ARG((workspace || saved) && us3 && B > 0 && F > 0 && S > 0);
    const size_t n = ts_units(B, F, S);
    unsigned long long* t = (unsigned long long*)malloc(n * 16);
    if (!t) return HM_ERR_LAUNCH;
    const void* src = saved ? saved : (const void*)carve((void*)workspace, B, V, F, S).ts;
    if (hipMemcpyAsync(t, src, n * 16, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        free(t);
        return HM_ERR_LAUNCH;
    }
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const size_t lo[4] = {0, ts_raster_units(B, S), ts_raster_units(B, S) + ts_lines_units(B, F, S), n};
    for (int k = 0; k < 3; ++k) {
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (size_t u = lo[k]; u < lo[k + 1]; ++u) {
            if (t[2 * u] != 0ull && t[2 * u] < t0) t0 = t[2 * u];
            if (t[2 * u + 1] > t1) t1 = t[2 * u + 1];
        }
        us3[k] = (t0 != ~0ull && t1 > t0) ? (float)((double)(t1 - t0) * 1e3 / (double)khz) : 0.f;
    }
    free(t);
    return HM_OK;
}
// durations (ms) of the LAST timed k_raster_fwd, k_bwd_lines, k_bwd_sweep launches -> ms3 (HOST pointer).  Waits for them.
int hm_debug_sil_timing_read(float* ms3)
{
    HM_CHECK_ARG(ms3 && g_timing);
    if (hipEventSynchronize(g_tev[4]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3, g_tev[0], g_tev[1]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3 + 1, g_tev[2], g_tev[3]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3 + 2, g_tev[3], g_tev[4]) != hipSuccess) return HM_ERR_LAUNCH;
    return HM_OK;
}
int hm_debug_read_partials(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.partials, (size_t)B * (S / 8) * (S / 8) * 16, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
// test hook: cap > 0 shrinks the unit table and the partial-slot table of the sweep work list to `cap` entries (the
// workspace keeps its size), so that small inputs exercise the beyond-capacity paths; 0 restores the defaults.
int hm_debug_sweep_caps(int cap)
{
    const int prev = hm_sweep_cap_override();
    hm_sweep_cap_override() = cap > 0 ? cap : 0;
    return prev;
}
int hm_debug_occupancy(int* raster_fwd_blocks, int* sweep_blocks)
{
    const int e1 = hm_raster_fwd_occupancy(raster_fwd_blocks), e2 = hm_sweep_occupancy(sweep_blocks);
    return (e1 == HM_OK && e2 == HM_OK) ? HM_OK : HM_ERR_LAUNCH;
}

// debug / test access to forward intermediates held in the workspace
int hm_sil_read_idx_map(const void* workspace, int B, int V, int F, int S, int* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.idx_map, (size_t)B * 4 * S * S * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
int hm_sil_read_boxes(const void* workspace, int B, int V, int F, int S, void* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.boxes, (size_t)B * F * 8, hipMemcpyDeviceToDevice, stream) == hipSuccess ? HM_OK
                                                                                                         : HM_ERR_LAUNCH;
}
int hm_sil_read_faces9(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.faces9, (size_t)B * F * 9 * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
// hm_sil_fwd with persistent_outputs skips the epilogue of an empty region whose outputs already hold the empty pattern - which
// depends on the loss inputs (keep / ref).  A caller that REUSES a workspace and its output buffers for another clip (new
// masks in the same buffers) calls this once after loading them: every region writes its outputs again on the next forward.
int hm_sil_invalidate_outputs(void* workspace, int B, int V, int F, int S, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    return hipMemsetAsync(w.region_state, 0, (size_t)B * (S / 16) * (S / 16), stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}
// (B,F,3,2) doubles: the per-(face, corner) sums of the last backward (tests: compared bit for bit with the CPU oracle)
int hm_sil_read_parts(const void* workspace, int B, int V, int F, int S, double* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.parts, (size_t)B * F * 6 * 8, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
}  // extern "C"
