/*
 * homan_amd.h -- C ABI of libhoman_amd.so: the MI355X-native leaves of HOMan's joint-optimisation hot path.
 *
 * The reference (hassony2/homan) is pure Python; the native code on its hot path lives in third-party CUDA /
 * PyTorch extensions (`neural_renderer`, `sdf`, `mano`) bound through pybind torch extensions.  This header is the
 * drop-in boundary for that native layer: plain C, device pointers + sizes, no torch types.  Each entry point names
 * the reference call site (file:line under /root/reference) whose arithmetic it carries.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, gfx950) unless stated otherwise; tensors are dense, row-major, fp32 /
 *     int32; the caller owns every buffer (outputs and workspaces), the library allocates nothing;
 *   - all work is enqueued on `stream` and is asynchronous w.r.t. the host (capturable in a hipGraph), except
 *     hm_bench_raster_fwd and the hm_debug_* helpers;
 *   - return value: HM_OK (0) or a negative error code; nothing throws across the ABI;
 *   - re-entrant per (workspace, stream).  State outside the caller's buffers: the hm_tune_* LAUNCH HINTS (grid sizes / LDS
 *     ballast of a few kernels: scheduling only, never results), which are PER CALLING THREAD (thread-local) and read when an
 *     entry point is called or captured - a thread sets them, issues or captures its launches, restores them; threads do not
 *     see each other's values; and the process-wide hm_debug_* hooks (timing events, capacity overrides: test tools);
 *   - workspaces that hold a reduction ticket (hm_reduce_workspace_bytes, hm_sil_workspace_bytes,
 *     hm_collision_workspace_bytes) must be zero-filled ONCE before first use; the ticket resets itself.
 *
 * `hipStream_t` is declared as an opaque pointer so that the header is usable from plain C hosts (ctypes, cgo...).
 */
#ifndef HOMAN_AMD_H
#define HOMAN_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef HIP_INCLUDE_HIP_HIP_RUNTIME_API_H
typedef struct ihipStream_t* hipStream_t;
#endif

#define HM_OK 0
#define HM_ERR_BAD_ARG (-1)
#define HM_ERR_LAUNCH (-2)
#define HM_ERR_UNSUPPORTED (-3)

/* ------------------------------------------------------------------ rigid transforms
 * reference homan/utils/geometry.py:9-27 (rot6d_to_matrix) + homan/utils/camera.py:108-139
 * (compute_transformation_persp), called from homan/homan.py:298-307 (object) and :341-382 (hand).
 *   verts[n,v,:] = (s * mesh[n,v,:]) @ R(rot6d[n]) + trans[n]      s = |scale[0]| if abs_scale else scale[0]
 * mesh (N,V,3), rot6d (N,3,2), trans (N,3), scale (1), rotmat (N,3,3) optional output, verts (N,V,3). */
int hm_rigid_fwd(const float* mesh, const float* rot6d, const float* trans, const float* scale, int abs_scale, int N,
                 int V, float* rotmat, float* verts, hipStream_t stream);
/* g_terms / weights: HOST arrays of n_terms (<= 5) device pointers (N,V,3) and factors, read at launch; their weighted sum
 * is d/dverts reaching mesh, scale, R, t (NULL entries are skipped).  g_rigid (N,V,3) and g_frame (one 3-vector per
 * frame at g_frame + n*frame_stride, times frame_scale, applied to every vertex) reach R, t only: gradients w.r.t. the
 * mesh-detached twin of the vertices.  Outputs: g_mesh (N,V,3) optional, g_rot6d (N,3,2), g_trans (N,3), g_scale_part (N)
 * optional (sum = d/dscale). */
int hm_rigid_bwd(const float* mesh, const float* rot6d, const float* scale, int abs_scale, const float* const* g_terms,
                 const float* weights, int n_terms, const float* g_rigid, const float* g_frame, int frame_stride,
                 float frame_scale, int N, int V, float* g_mesh, float* g_rot6d, float* g_trans, float* g_scale_part,
                 void* workspace, hipStream_t stream);
/* workspace of hm_rigid_bwd / hm_rigid_bwd_sil (zero-filled once; per-frame tickets reset themselves): with it the frame
 * is split over ceil(V/256) workgroups and the last one finishes; NULL = one workgroup per frame. */
size_t hm_rigid_workspace_bytes(int N);
/* Scheduling hint, no effect on results (exact sums): 1 = hm_rigid_bwd_sil* as ceil(V / 256) small workgroups per frame + a
 * per-frame ticket instead of one large workgroup per frame.  Per calling thread (thread-local), read at launch / capture; returns the previous
 * value; < 0 only queries. */
int hm_tune_rigid_chunked(int enable);
/* hm_rigid_bwd with the silhouette gradient as one more full term, gathered on the fly from the per-(face, corner) NDC
 * gradients of the edge sweeps (sil_parts = hm_sil_parts(workspace) after an hm_sil_bwd called with grad_verts == NULL;
 * adj_off / adj_items / cam_verts / K / orig_size / F as given to that call): no gather launch, no (N,V,3) round trip. */
int hm_rigid_bwd_sil(const float* mesh, const float* rot6d, const float* scale, int abs_scale, const float* const* g_terms,
                     const float* weights, int n_terms, const double* sil_parts, const int* adj_off, const int* adj_items,
                     const float* cam_verts, const float* K, float orig_size, int F, int N, int V, float* g_rot6d,
                     float* g_trans, float* g_scale_part, void* workspace, int sum_log2q, hipStream_t stream);
/* ORDER-INDEPENDENT SUMS (sum_log2q).  Every reduction on the object's gradient chain - the per-(face, corner) sums of the
 * edge sweeps, the vertex gather, the per-frame sums of the rigid backward - rounds its addends to multiples of the quantum
 * 2^sum_log2q (0 = the default 2^-44; -60 <= sum_log2q <= 0) and adds them in double, which is exact while |sum| < 2^53
 * quanta: the result is a function of the SET of addends, independent of launch geometry, unit composition and atomics'
 * arrival order, and the CPU oracle (oracle/csrc/objchain.c) reproduces it bit for bit.  With the per-term arithmetic in
 * IEEE operations this makes the free-running object trajectory of reference homan/jointopt.py:158-192 bit-equal to the
 * CPU path's.  Callers whose gradients are O(1) or larger (pose initialisation: unnormalised sums of squares) pass a
 * coarser grid, e.g. -24; both calls of a backward (hm_sil_bwd*, hm_rigid_bwd_sil*) must be given the same value. */
/* Off-screen penalty of the object-pose initialisation, reference homan/pose_optimization.py:112-135: hinge on the six
 * clipping planes of the projected vertices (K: ONE (3,3) camera, normalised, orig_size 1), per candidate pose:
 * out[n] = weight * sum_v (...), grad (N,V,3) = d out[n] / d verts[n]. */
int hm_offscreen_fwd(const float* verts, const float* K, int N, int V, float zfar, float weight, float* out, float* grad,
                     hipStream_t stream);
/* Best-ever bookkeeping of the pose initialisation's loop, reference homan/pose_optimization.py:340-353, in one launch:
 * losses_out[i] = sums[i * stride] + extra[i]; if the first minimum (torch.argmin) is < best_loss[0], candidate ind's CURRENT
 * rot6d (6) / trans (3) and the minimum are copied to best_rot6d / best_trans / best_loss (one float, start it at +inf). */
int hm_pose_keep_best(const float* sums, int stride, const float* extra, int n, const float* rot6d, const float* trans,
                      float* best_loss, float* best_rot6d, float* best_trans, float* losses_out, hipStream_t stream);
/* The same evaluation WITHOUT touching a best-ever state, for a caller that walks the candidates of one fit as several independent
 * loops (own Adam state and step counter each, their launches overlapping on separate streams) and applies the rule of :340-353
 * over all of them afterwards: row step[0] - 1 of log (max_steps, 16) receives {minimum of this loop's candidates, its index
 * (int bits; -1: none), 1.0 if some loss is NaN, that candidate's CURRENT rot6d (6) and trans (3), 0 x 4}; step = the device
 * counter of hm_adam_step, read after the step that followed the evaluation. */
int hm_pose_keep_best_log(const float* sums, int stride, const float* extra, int n, const float* rot6d, const float* trans,
                          const int* step, int max_steps, float* log, float* losses_out, hipStream_t stream);
/* out = s[0] * in ;  out = s0[0]*a + s1[0]*b   (backward of the losses whose unit gradient is produced forward) */
int hm_scale_by(const float* in, const float* s, long n, float* out, hipStream_t stream);
int hm_scale2_by(const float* a, const float* s0, const float* b, const float* s1, long n, float* out,
                 hipStream_t stream);
/* out = w0*a0 + w1*a1 + w2*a2 + w3*a3 (NULL terms skipped): the weighted sum of per-loss vertex gradients, i.e. the
 * `loss = sum_k lw_k * loss_k` weighting of reference homan/jointopt.py:180-188 applied to gradients. */
int hm_lincomb4(const float* a0, float w0, const float* a1, float w1, const float* a2, float w2, const float* a3,
                float w3, long n, float* out, hipStream_t stream);
/* out[0] = w0 * sum(parts[0..n)) + w1 * extra[0]  (extra may be NULL) */
int hm_sum_small(const float* parts, int n, float w0, const float* extra, float w1, float* out, hipStream_t stream);

/* ------------------------------------------------------------------ MANO linear blend skinning
 * reference homan/manomodel.py:84-151 (ManoModel.forward_pca, right hand) + the `mano` layer it calls
 * (manomodel.py:119-123) + "+ mano_trans" of homan/homan.py:356.
 * model: host array of 8 device pointers {v_template (778,3), M (145,2334) = [posedirs ; shapedirs^T],
 *   J_template (16,3), J_shapedirs (16,3,10), lbs_weights (778,16), pca components (16,45), hand_mean (45),
 *   parents (16) int32}.   pca (B,pca_dim>=16; the first 16 columns drive the mesh), rot (B,3), betas (B,10),
 *   trans (B,3) or NULL.  verts (B,778,3); joints (B,16,3) optional.
 *   verts_world (B,778,3) optional: the rigid hand transform of hm_rigid_fwd (rigid_rot6d (B,3,2), rigid_trans (B,3),
 *   rigid_scale (1), no abs) applied in the same launch.
 *   state (hm_mano_state_bytes(B)) optional: kinematic-chain state + posed vertices kept for hm_mano_bwd. */
int hm_mano_fwd(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d, const float* rigid_trans,
                const float* rigid_scale, float* verts_world, float* state, hipStream_t stream);
size_t hm_mano_workspace_bytes(int B);      /* zero-filled once by the caller (self-resetting per-frame tickets) */
size_t hm_mano_state_bytes(int B);
/* g_pca_extra (B,pca_dim) optional: g_pca = d/d pca through the mesh + w_extra * g_pca_extra (e.g. the PCA prior).
 * state: what hm_mano_fwd stored for the SAME parameters, or NULL (the chain is then recomputed). */
int hm_mano_bwd(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                const float* g_verts, const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                float* g_trans, const float* state, void* workspace, hipStream_t stream);
/* hm_rigid_bwd_clips of the hand + hm_mano_bwd in ONE launch (no mesh-gradient buffer in between): mesh = the forward's
 * model-space vertices (B,778,3), rigid_rot6d / rigid_scale = the hand's rigid pose (scale: one per clip of clip_len frames),
 * g_terms .. frame_scale as hm_rigid_bwd_clips, g_rigid_rot6d (B,3,2) / g_rigid_trans (B,3) receive the rigid pose's gradients.
 * Replaces the hand branch of loss.backward() through reference homan/homan.py:341-382 (rigid transform) and
 * homan/manomodel.py:84-151 (LBS).  Same expressions as the two launches; the sums of dR / dt are formed per vertex chunk. */
int hm_mano_bwd_rigid_clips(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                            const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                            float* g_trans, const float* state, void* workspace, const float* mesh, const float* rigid_rot6d,
                            const float* rigid_scale, const float* const* g_terms, const float* weights, int n_terms,
                            const float* g_rigid, const float* g_frame, int frame_stride, float frame_scale,
                            float* g_rigid_rot6d, float* g_rigid_trans, int clip_len, hipStream_t stream);

/* ------------------------------------------------------------------ silhouette rasteriser + fused masked-MSE / IoU
 * reference homan/losses.py:183-197 (compute_sil_loss_object) and the `neural_renderer` call inside it
 * (losses.py:187; renderer built at losses.py:73-77): projection with the ROI intrinsics K (orig_size), fill_back,
 * hard z-buffer raster at 2S x 2S samples, vertical flip, 2x2 average pool, NMR edge-sweep pseudo-gradient.
 *   verts (B,V,3) camera space; faces (F,3) int32 shared by all frames (faces_bstride = 0) or (B,F,3) (= 3F);
 *   K (B,3,3); pooled (B,S,S) silhouettes out.
 *   Fused loss (all four non-NULL): keep/ref (B,S,S), keep_sum (1) = sum(keep);
 *     loss_out[0] = sum((keep*sil - ref)^2) / keep_sum / B ; loss_out[1] = mean_b IoU_b   (losses.py:188-196).
 *   work_order: B*(S/16)^2 int32 entries (frame << 16 | region) = dispatch order of the (frame, 32x32-sample region)
 *     workgroups (a permutation; expensive ones first), or NULL for frame-major order.
 *   pooled_depth (B,S,S) optional: the depth image nr.Renderer.render returns beside the silhouette (reference
 *     homan/homan.py:391,406): z-buffer (zfar where empty), flipped, 2x2 average pooled.
 *   alpha_full (B,2S,2S) optional: the un-pooled coverage image, vertically flipped like the output = what
 *     nr.Renderer(image_size=2S, anti_aliasing=False) returns (reference homan/pose_optimization.py:89-96); its
 *     backward is hm_sil_bwd mode 3.  With keep / ref given as well, they are (2S,2S)-resolution images and the fused loss is
 *     per SAMPLE (reference homan/pose_optimization.py:140-143); per-frame sums come from hm_sil_reduce(frame_out), the
 *     backward is hm_sil_bwd mode 4.
 *   mask_shared: bit 0 - keep / ref have no batch dimension (one mask for every frame); bit 1 (with (2S,2S) keep / ref) - the
 *     per-sample loss WITHOUT its per-sample outputs: alpha_full may be NULL, nothing of (B,2S,2S) size is written; the masks
 *     must be binary (0 / 1) and the backward is hm_sil_bwd mode 5 (the pose initialisation's loop: 500 candidates write
 *     260 MB less per step).
 *   rigid_rot6d (B,3,2) / rigid_trans (B,3) / rigid_scale (1) / rigid_abs optional: `verts` are then mesh-space and the
 *     rigid transform of hm_rigid_fwd is applied in the face-setup kernel (same arithmetic), so the silhouette chain does
 *     not wait for a separate transform launch; hm_sil_bwd still takes the camera-space vertices.
 *   persistent_outputs != 0: the caller passes the SAME pooled / pooled_depth / keep / ref buffers as in the previous call on
 *     this workspace (a fixed optimisation loop); regions that were background then and are background now are not
 *     rewritten.  0 = every output is written.
 *   S must be a multiple of 16 (32 and <= 512 for the silhouette backward). */
size_t hm_sil_workspace_bytes(int B, int V, int F, int S);
int hm_sil_fwd(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
               float orig_size, float znear, float zfar, const float* keep, const float* ref,
               const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
               float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
               const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, hipStream_t stream);
/* scheduling hint (no reference counterpart, no effect on results): winding class (0: faces as stored, 1: the reversed
 * copies of fill_back) that holds the camera-facing surface of the mesh rendered on this workspace.  hm_sil_fwd rasterises
 * it first and rejects the units of the other class against per-block hidden depths before any per-sample work. */
int hm_sil_hint_near_winding(void* workspace, int winding, hipStream_t stream);
/* deferred loss / IoU reduction of an hm_sil_fwd called with keep/ref but loss_out == NULL (off the critical path) */
int hm_sil_reduce(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out, void* workspace,
                  hipStream_t stream);       /* frame_out (B,2) optional: per-frame {sum of squares, IoU}; loss_out may be NULL then */
/* mode 3: grad_pooled is (B,2S,2S) = dL/d alpha_full (rendering without anti-aliasing).
 * mode 4: fused per-sample L2 without anti-aliasing: upstream (B) = dL/d(per-frame sums of squares), all > 0.
 * mode 5: mode 4 for BINARY keep / ref masks: keep (keep alpha - ref) is -1 at every uncovered sample that pulls and +1 at every
 *   covered one that pushes, so the line pass reads no per-sample gradient (same results as mode 4 on such masks, bit for bit).
 * mode 1: upstream (1) = dL/d loss_out[0]; mode 2: same with upstream[0] > 0 guaranteed by the caller (the forward's
 * sweep planes are reused, one launch less); mode 0: grad_pooled (B,S,S) = dL/d pooled.  adj_off (V+1), adj_items (3F):
 * CSR vertex -> (face*3 + corner).  face_order: B*F int32 permutation of frame*F+face (visiting order of the edge
 * sweeps, expensive faces first) or NULL.  grad_verts (B,V,3) overwritten, or NULL to skip the vertex gather (the
 * per-corner gradients stay in the workspace: hm_sil_parts / hm_rigid_bwd_sil); grad_ndc (B,V,3) optional. */
int hm_sil_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
               const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
               const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
               int sum_log2q, hipStream_t stream);
/* Backward of the depth image (neural_renderer backward_depth_map, reached from reference homan/homan.py:391,406):
 * grad_pooled_depth (B,S,S) -> grad_verts (B,V,3), for the frame state the last hm_sil_fwd left in `workspace`. */
int hm_depth_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size,
                 const float* grad_pooled_depth, const int* adj_off, const int* adj_items, float* grad_verts,
                 void* workspace, hipStream_t stream);
/* The same with the NON-ZERO STRUCTURE of the upstream image handed over (gflags, optional; S % 64 == 0): B * S * (S / 64) bytes,
 * one per (frame, pixel row, 64-pixel segment), non-zero wherever some pixel of the segment has a non-zero gradient (as
 * hm_ordinal_depth_bwd_flags writes them; a set byte over an all-zero segment is harmless).  An ordinal depth term is zero
 * wherever render and annotation agree on the order - nearly everywhere -, and faces / frames that touch no flagged segment
 * get their exact zeros without being walked.  Same result as hm_depth_bwd. */
int hm_depth_bwd_sparse(const float* verts, const float* K, int B, int V, int F, int S, float orig_size,
                        const float* grad_pooled_depth, const int* adj_off, const int* adj_items, float* grad_verts,
                        const unsigned char* gflags, void* workspace, hipStream_t stream);
/* Ordinal depth loss between two rendered layers (0 = object, 1 = hand): reference homan/homan.py:384-419 +
 * homan/lossutils.py:133-169 (as the method intends; the reference call site raises before reaching it, DESIGN.md).
 * d*/a*: depth / silhouette renders (B,S,S) f32; m*: instance masks (B,S,S) u8.  frame_part: B*8 floats (8-byte aligned),
 * ZERO-filled once by the caller (per-frame integer records of the chunk workgroups; the call leaves them zero again);
 * rec: 5 floats {num_pairs, n01, sum01, n10, sum10} kept for the backward; out1[0] = loss.
 * workspace: the reduce workspace (see the small losses below), zero-filled once. */
int hm_ordinal_depth_fwd(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                         const unsigned char* m1, int B, int S, float* frame_part, float* rec, float* out1,
                         void* workspace, hipStream_t stream);
int hm_ordinal_depth_bwd(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                         const unsigned char* m1, int B, int S, const float* rec, const float* upstream, float* g0,
                         float* g1, hipStream_t stream);
/* ... and the per-segment non-zero flags of both gradient images (flags0 / flags1, see hm_depth_bwd_sparse; both or neither) */
int hm_ordinal_depth_bwd_flags(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                               const unsigned char* m1, int B, int S, const float* rec, const float* upstream, float* g0,
                               float* g1, unsigned char* flags0, unsigned char* flags1, hipStream_t stream);
/* scheduling hint (no reference counterpart, no effect on results): persistent workgroups of the edge-sweep kernel,
 * default 1280; 768 suits loops whose other streams carry the longer chain (collision + contact terms).  Per calling thread (thread-local),
 * read when hm_sil_bwd is called or captured.  Returns the previous value; blocks <= 0 only queries. */
int hm_tune_sweep_blocks(int blocks);
/* Scheduling hint, no effect on results: bytes of unused dynamic LDS added to every rasteriser launch (3072 caps a CU at 5
 * rasteriser workgroups instead of 6, which leaves registers / LDS for the kernels of the caller's other stream).  Per calling thread (thread-local),
 * read when hm_sil_fwd is called (or captured).  Returns the previous value; bytes < 0 only queries. */
int hm_tune_raster_lds_pad(int bytes);
/* Scheduling hint, no effect on results: adaptive launch order of the rasteriser's workgroups.  While on, the forward launches
 * record the time every workgroup took and the backward's first launch re-sorts the order, longest first, for the next forward
 * of the same workspace (the `work_order` argument only seeds it): what is expensive moves during a fit.  enable > 0 / 0 / < 0
 * (query).  Per calling thread (thread-local), read when hm_sil_fwd / hm_sil_bwd are called (or captured).  Returns the previous value. */
int hm_tune_raster_reorder(int enable);
/* Same for the metric-only nearest-vertex search (small latency-bound workgroups that otherwise take every wave slot of a CU
 * next to the kernel they overlap): 65536 = two search workgroups per CU. */
int hm_tune_nn_lds_pad(int bytes);
/* The same ballast for the other kernels of the hand side, by family: 0 MANO forward, 1 MANO backward, 2 the smoothness /
 * interaction / hand-terms launches, 3 the fused pair-terms launch, 4 rigid backward.  Returns the previous value (-1: no such
 * family); bytes < 0 only queries. */
int hm_tune_lds_pad(int family, int bytes);
/* test hook: cap > 0 shrinks the capacity tables of the sweep work list so that small inputs take the beyond-capacity
 * paths (binary search for a unit's first face, atomically accumulated faces); 0 restores the defaults.  Returns the
 * previous value. */
int hm_debug_sweep_caps(int cap);
/* rgb output of nr.renderer.Renderer.render for the reference's texture_size-1 per-face colours (reference
 * homan/homan.py:535-538 render_limem, light set at :173-176, colours from homan/meshutils.py:7-51): (B,3,S,S) image of
 * the LAST hm_sil_fwd on this workspace (same verts / faces), flat lighting ambient + directional * relu(<n, dir>) on the
 * camera-space face normals, background where no face covers, vertical flip and 2x2 average as the other outputs.
 * textures (B,F,3); light_dir, background: HOST float[3]. */
int hm_shade_rgb(const float* verts, const int* faces, int faces_bstride, const float* textures, int B, int V, int F, int S,
                 const float* light_dir, float intensity_ambient, float intensity_directional, const float* background,
                 float* rgb, void* workspace, hipStream_t stream);
/* device pointer to the (B,F,3,2) DOUBLES left in `workspace` by the last hm_sil_bwd: per-(face, corner) NDC gradients as
 * exact sums on the grid 2^sum_log2q */
const double* hm_sil_parts(const void* workspace, int B, int V, int F, int S);
/* forward intermediates kept in the workspace (tests): face-index map (B,2S,2S) int32, packed NDC faces (B,F,9) */
int hm_sil_read_idx_map(const void* workspace, int B, int V, int F, int S, int* out, hipStream_t stream);
int hm_sil_read_faces9(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream);
/* hm_sil_fwd(persistent_outputs = 1) leaves the epilogue of an empty region out when its outputs already hold the empty
 * pattern, which depends on keep / ref.  After loading ANOTHER clip's masks into the same buffers (a resident stepper fitting
 * a stream of clips, reference fit_vid_dataset.py:190-379) call this once: the next forward writes every region again. */
int hm_sil_invalidate_outputs(void* workspace, int B, int V, int F, int S, hipStream_t stream);
/* the (B,F,3,2) doubles of hm_sil_parts copied to a caller's device buffer (tests) */
int hm_sil_read_parts(const void* workspace, int B, int V, int F, int S, double* out, hipStream_t stream);
/* per-face screen boxes (B,F) x 8 bytes {x0|winding<<14, y0, x1, y1} u16 */
int hm_sil_read_boxes(const void* workspace, int B, int V, int F, int S, void* out, hipStream_t stream);

/* ------------------------------------------------------------------ small losses (value + unit gradient in one launch)
 * workspace for all of them: hm_reduce_workspace_bytes(), zero-filled once. */
size_t hm_reduce_workspace_bytes(void);
/* reference homan/losses.py:141-164: out2[0] = mean sum_xy (proj - ref/image_size)^2, out2[1] = mean px distance */
int hm_v2d_fwd(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
               int V, float* unit_grad, float* out2, void* workspace, hipStream_t stream);
/* reference homan/lossutils.py:18-36: temporal smoothness of verts (N,V,3), frames interleaved by hand_nb */
int hm_smooth_fwd(const float* verts, int N, int V, int hand_nb, float* unit_grad, float* out1, void* workspace,
                  hipStream_t stream);
/* reference homan/lossutils.py:39-40 and :107-109: out3 = {mean(pca^2), (s_obj-m_obj)^2, (s_hand-m_hand)^2} */
int hm_priors_fwd(const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                  const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out3, hipStream_t stream);
/* hm_v2d_fwd + hm_smooth_fwd (+ hm_priors_fwd when pca != NULL) of the hand vertices in ONE launch (same outputs) */
int hm_hand_terms_fwd(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
                      int V, float* unit_v2d, float* out_v2d2, float* unit_smooth, float* out_smooth1, const float* pca,
                      long npca, const float* s_obj, const float* m_obj, const float* s_hand, const float* m_hand,
                      float* g_pca, float* g_sobj, float* g_shand, float* out_priors3, void* workspace, hipStream_t stream);
/* reference homan/losses.py:199-242 ('centroid') with the gating of :98-139 (project_bbox :20-49, compute_iou
 * utils/bbox.py:111-135, compute_dist_z utils/geometry.py:69-86).  out1 = un-normalised sum; frame_rec (B,8). */
int hm_inter_fwd(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                 float expansion, float zthresh, float* frame_rec, float* out1, void* workspace, hipStream_t stream);
int hm_inter_bwd(const float* frame_rec, const float* upstream, int B, int Vh, int Vo, float* g_hand, float* g_obj,
                 hipStream_t stream);

/* ------------------------------------------------------------------ contact (Chamfer direction hand -> object)
 * reference homan/interactions/contactloss.py:60-79,162-163 (pairwise distances, arg-min over the object) and the
 * metric of homan/losses.py:225-241: metric_out[0] = max_b sqrt(min_ij |h_i - o_j|^2).  nn_idx = nn_d2 = NULL: metric only
 * (the same exact value; object-vertex groups that cannot hold the minimum are skipped by a bounding-sphere test). */
int hm_nn_fwd(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
              float* metric_out, void* workspace, hipStream_t stream);
/* reference homan/lossutils.py:112-130 -> contactloss.py:149-309 as executed: out1 = mean thresh*tanh(|nn-h|/thresh) */
int hm_contact_fwd(const float* verts_hand, const float* verts_obj, const int* nn_idx, int B, int Vh, int Vo,
                   float thresh, float* g_hand, float* g_obj, float* out1, void* workspace, hipStream_t stream);

/* ------------------------------------------------------------------ SDF interpenetration
 * reference homan/lossutils.py:43-64 -> homan/interactions/scenesdf.py:77-148 and the `sdf` package (scenesdf.py:119).
 * Scene = {0: hand (closed faces), 1: object}.  out1[0] = sum of grid_sample(clamp(SDF_k,0), verts_l) over both
 * ordered pairs and all frames; g0 / g1 = d out / d verts0 / d verts1. */
size_t hm_collision_workspace_bytes(int B, int V0, int V1, int F0, int F1);
int hm_collision_fwd(const float* verts0, const int* faces0, int V0, int F0, const float* verts1, const int* faces1,
                     int V1, int F1, int B, float scale_factor, float* g0, float* g1, float* out1, void* workspace,
                     hipStream_t stream);
/* per-vertex penetration depths in world units (reference homan/interactions/scenesdf.py:141-146 `dist_values`, read by
 * homan/eval/pointmetrics.py:102-124): dv0 (B,V0) = clamp(SDF of mesh 1, 0) at the vertices of mesh 0 = dist_values[(1,0)],
 * dv1 (B,V1) = dist_values[(0,1)]; from the workspace of the last hm_collision_fwd on the same verts0 / verts1. */
int hm_collision_dist_values(const float* verts0, int V0, const float* verts1, int V1, int F0, int F1, int B, float* dv0,
                             float* dv1, void* workspace, hipStream_t stream);
/* clamp(SDF,0) on the full 32^3 grid of mesh `which`, from the workspace of the last hm_collision_fwd */
int hm_collision_read_grid(const int* faces, int V, int F, int B, int which, int V0, int V1, int F0, int F1, float* phi,
                           void* workspace, hipStream_t stream);

/* ------------------------------------------------------------------ optimiser step + logging
 * reference homan/jointopt.py:138-151,192 (torch.optim.Adam, three groups) and :184-189 (loss_evolution).
 * slots: n_tensors records {float* p, g, m, v; long n; float lr; int pad} (hm_adam_slot_bytes() each);
 * step: TWO device int32 words {completed steps (incremented by the launch), ticket word (zero between launches)};
 * zero_grad != 0 clears g after the update. */
size_t hm_adam_slot_bytes(void);
int hm_adam_step(const void* slots, int n_tensors, int* step, float beta1, float beta2, float eps, int zero_grad,
                 int blocks_per_tensor, hipStream_t stream);
/* hm_log_total_clips + hm_adam_step in one launch: the log row of the step being taken (weighted totals + every slot of
 * `vals`) is written by an extra grid row before the device step counter moves. */
int hm_adam_step_log(const void* slots, int n_tensors, int* step, float beta1, float beta2, float eps, int zero_grad,
                     int blocks_per_tensor, float* vals, const float* weights, int n, int max_steps, float* log, int nclips,
                     hipStream_t stream);
/* vals[n] = sum_i weights[i]*vals[i] (the weighted total of jointopt.py:180-188), then log row step[0] = vals[0..n] */
int hm_log_total(float* vals, const float* weights, int n, const int* step, int max_steps, float* log,
                 hipStream_t stream);
/* log[step[0]*n + i] = src[i] */
int hm_log_scalars(const float* src, int n, const int* step, int max_steps, float* log, hipStream_t stream);

/* ------------------------------------------------------------------ clip batches: C clips, ONE launch per kernel
 * reference: clips are independent optimisations (one HOMan + one Adam per clip, homan/jointopt.py:92-151); the only
 * sharding hook upstream is the strided sample selection of fit_vid_dataset.py:54-55,190.  Frames of one clip stay
 * coupled through the smoothness term (homan/lossutils.py:18-36) and the per-clip normalisers (sum(keep) and 1/B of
 * homan/losses.py:189-194, the means of losses.py:158 / lossutils.py:39-40 / contactloss.py:284-285, the per-clip
 * sums of losses.py:233-239 and scenesdf.py:147), so a batch concatenates whole clips along the frame axis:
 *   - N (or B) = clips * clip_len frames, clip c = frames [c*clip_len, (c+1)*clip_len); clip_len == 0: one clip;
 *   - every per-clip scalar input (scale, s_obj, m_obj, s_hand, m_hand, keep_sum, rigid_scale) is an array with one
 *     entry per clip; every per-clip scalar output is written at out + c*out_stride (+ the offsets of the plain call);
 *   - reduce workspaces hold `clips` slices of hm_reduce_workspace_bytes() back to back (zero-filled once);
 *   - the clips must share V, F, S and the face topology; vertices, masks, cameras are per frame as before.
 * Each clip's sums are formed by its own workgroups in the order of a single-clip launch: a batched call returns, per
 * clip, bit-identical results to the plain entry point called on that clip alone (tests/test_clip_batch_gpu.py).
 * The plain entry points above are these with clip_len = 0, out_stride = 0. */
int hm_rigid_fwd_clips(const float* mesh, const float* rot6d, const float* trans, const float* scale, int abs_scale, int N,
                       int V, float* rotmat, float* verts, int clip_len, hipStream_t stream);
int hm_rigid_bwd_clips(const float* mesh, const float* rot6d, const float* scale, int abs_scale,
                       const float* const* g_terms, const float* weights, int n_terms, const float* g_rigid,
                       const float* g_frame, int frame_stride, float frame_scale, int N, int V, float* g_mesh,
                       float* g_rot6d, float* g_trans, float* g_scale_part, void* workspace, int clip_len,
                       hipStream_t stream);
int hm_rigid_bwd_sil_clips(const float* mesh, const float* rot6d, const float* scale, int abs_scale,
                           const float* const* g_terms, const float* weights, int n_terms, const double* sil_parts,
                           const int* adj_off, const int* adj_items, const float* cam_verts, const float* K,
                           float orig_size, int F, int N, int V, float* g_rot6d, float* g_trans, float* g_scale_part,
                           void* workspace, int clip_len, int sum_log2q, const float* smooth_verts, float smooth_weight,
                           hipStream_t stream);
/*   smooth_verts (optional): camera-space vertices (N,V,3) of the same mesh; the gradient of smooth_weight * the temporal
 *   smoothness term of every clip (reference homan/lossutils.py:18-36) is formed inside the launch and added before g_terms -
 *   the floats hm_smooth_fwd_clips' unit gradient times the weight gives as a first term, without waiting for that launch. */
/* out[c] = w0 * sum(parts[c*n .. c*n+n)) + w1 * extra[c] */
int hm_sum_small_clips(const float* parts, int n, float w0, const float* extra, float w1, float* out, int nclips,
                       hipStream_t stream);
int hm_mano_fwd_clips(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                      const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d,
                      const float* rigid_trans, const float* rigid_scale, float* verts_world, float* state, int clip_len,
                      hipStream_t stream);
/* Two hands per frame arrive interleaved frame-major [h0_t0, h1_t0, h0_t1, ...] and hand i is the strided slice i::hand_nb of
 * every MANO parameter, through the MANO model of ITS side (reference homan/homan.py:62-63,343-358).  The *_rows entry points
 * evaluate such a slice in place: frame f of the launch is row row0 + f * row_stride of EVERY per-row array (parameters,
 * gradients, vertices, state; all sized for B * row_stride rows); clip_len counts rows. */
int hm_mano_fwd_rows(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas,
                     const float* trans, int B, float* verts, float* joints, const float* rigid_rot6d,
                     const float* rigid_trans, const float* rigid_scale, float* verts_world, float* state, int clip_len,
                     int row0, int row_stride, hipStream_t stream);
int hm_mano_bwd_rows(const void* const* model, const float* pca, int pca_dim, const float* rot, const float* betas, int B,
                     const float* g_verts, const float* g_pca_extra, float w_extra, float* g_pca, float* g_rot, float* g_betas,
                     float* g_trans, const float* state, void* workspace, int row0, int row_stride, hipStream_t stream);
int hm_sil_fwd_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                     float orig_size, float znear, float zfar, const float* keep, const float* ref,
                     const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                     float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                     const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                     int out_stride, float* cam_verts_out, hipStream_t stream);
/*   cam_verts_out (B,V,3) optional (with the rigid_* arguments): the camera-space vertices = what hm_rigid_fwd returns for
 *   the same inputs (same arithmetic, same floats), written by extra workgroups of the face-setup launch. */
/* The same forward in two calls, for a caller that forks a second stream off the camera-space vertices: phases = 1 launches
 * the face setup only (cam_verts_out is complete when it ends), 2 the rasteriser (+ reduction) only, 3 both; both calls take
 * the same arguments. */
int hm_sil_fwd_phase_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                           float orig_size, float znear, float zfar, const float* keep, const float* ref,
                           const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                           float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                           const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                           int out_stride, float* cam_verts_out, int phases, hipStream_t stream);
/* Several renders - their own meshes (V, F, vertices, faces), cameras, outputs and workspaces - as ONE face-setup launch and ONE
 * raster launch.  reference: the ordinal depth term renders the object twice per iteration (ROI silhouette homan/losses.py:187,
 * full-image depth homan/homan.py:391) and the hand once (homan.py:406); clips of a dataset come with their own meshes
 * (homan/datasets/core50.py:22-42) - a render per mesh is the per-frame mesh offset table of such a batch.  Every field means
 * what the argument of the same name means in hm_sil_fwd_clips (no alpha_full: anti-aliased renders only; the loss / IoU
 * reduction of a render with keep / ref: hm_sil_reduce_clips or the backward's loss_out).  The launch hints (hm_tune_raster_*)
 * apply as in hm_sil_fwd_clips: each render keeps its own (static or adaptive) order, render after render.
 * Results = n separate hm_sil_fwd_clips calls, bit for bit; each render's backward runs on its own workspace as before.
 * n <= 4; the workspaces must be distinct.  phases: 1 = face setup, 2 = raster, 3 = both. */
typedef struct HmSilRender {
    const float* verts; const int* faces; const float* K; const float* keep; const float* ref; float* pooled;
    const int* work_order; float* pooled_depth; const float* rigid_rot6d; const float* rigid_trans; const float* rigid_scale;
    float* cam_verts_out; void* workspace;
    int faces_bstride, B, V, F, S, mask_shared, rigid_abs, persistent_outputs, clip_len;
    float orig_size, znear, zfar;
} HmSilRender;
int hm_sil_fwd_multi(const HmSilRender* renders, int n, int phases, hipStream_t stream);
size_t hm_sil_render_bytes(void);      /* sizeof(HmSilRender) as the library was built: a binding checks its own layout against it */
int hm_sil_reduce_clips(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out,
                        void* workspace, int clip_len, int out_stride, hipStream_t stream);
int hm_sil_bwd_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                     const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                     const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                     int clip_len, float* loss_out, int out_stride, int sum_log2q, hipStream_t stream);
/* The same backward in two calls (phases: bit 0 = masks + line expansion + work list, bit 1 = edge sweeps + vertex gather; 3 =
 * hm_sil_bwd_clips), for a caller whose other streams wait for the END of the line expansion. */
int hm_sil_bwd_phase_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                     const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                     const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                     int clip_len, float* loss_out, int out_stride, int phases, int sum_log2q, hipStream_t stream);
/*   loss_out (optional, modes 1 / 2): the loss / IoU reduction of hm_sil_reduce_clips (same arithmetic) done by extra
 *   workgroups at the front of the backward's first launch, for a forward that was called with loss_out == NULL: the value
 *   is only logged, so it need not cost a launch on the chain raster -> lines -> sweeps. */
int hm_v2d_fwd_clips(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size, int N,
                     int V, float* unit_grad, float* out2, void* workspace, int clip_len, int out_stride,
                     hipStream_t stream);
int hm_smooth_fwd_clips(const float* verts, int N, int V, int hand_nb, float* unit_grad, float* out1, void* workspace,
                        int clip_len, int out_stride, hipStream_t stream);
/* npca = PCA entries of ONE clip (pca holds nclips * npca) */
int hm_priors_fwd_clips(const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                        const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out3, int nclips,
                        int out_stride, hipStream_t stream);
int hm_hand_terms_fwd_clips(const float* verts, const float* camintr, int hand_nb, const float* ref2d, float image_size,
                            int N, int V, float* unit_v2d, float* out_v2d2, float* unit_smooth, float* out_smooth1,
                            const float* pca, long npca, const float* s_obj, const float* m_obj, const float* s_hand,
                            const float* m_hand, float* g_pca, float* g_sobj, float* g_shand, float* out_priors3,
                            void* workspace, int clip_len, int out_stride, hipStream_t stream);
int hm_inter_fwd_clips(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                       float expansion, float zthresh, float* frame_rec, float* out1, void* workspace, int clip_len,
                       int out_stride, hipStream_t stream);
/* Up to three terms of the frames' (hand, object) vertex pairs in ONE launch (csrc/pairterms.hip); every term is optional
 * (output pointer NULL) and returns what its own entry point returns on the same inputs:
 *   metric_out -> hm_nn_fwd_clips with nn_idx = nn_d2 = NULL (<= 4096 object vertices)          reduce workspace ws_nn
 *   out_inter  -> hm_inter_fwd_clips (frame records `frame_rec`)                                 reduce workspace ws_inter
 *   out_smooth -> hm_smooth_fwd_clips(verts_obj, ..., hand_nb = 1) (unit gradient `unit_smooth`) reduce workspace ws_smooth
 *   ht_out_v2d2 -> hm_hand_terms_fwd_clips(verts_hand, camintr, hand_nb = 1, ht_...) (one hand per frame)  workspace ws_hand
 * The workspaces must be distinct (the terms run side by side).  Replaces, in one go, reference homan/losses.py:199-242
 * (loss + logged distance) and the object half of homan/lossutils.py:18-36. */
int hm_pair_terms_fwd_clips(const float* verts_hand, const float* verts_obj, const float* camintr, int B, int Vh, int Vo,
                            float* metric_out, const int* obj_order, void* ws_nn, float expansion, float zthresh,
                            float* frame_rec, float* out_inter, void* ws_inter, float* unit_smooth, float* out_smooth,
                            void* ws_smooth,
                            const float* ht_ref2d, float ht_image_size, float* ht_unit_v2d, float* ht_out_v2d2,
                            float* ht_unit_smooth, float* ht_out_smooth1, const float* ht_pca, long ht_npca,
                            const float* ht_s_obj, const float* ht_m_obj, const float* ht_s_hand, const float* ht_m_hand,
                            float* ht_g_pca, float* ht_g_sobj, float* ht_g_shand, float* ht_out_priors3, void* ws_hand,
                            const float* obj_spheres, const float* obj_rot6d, const float* obj_trans, const float* obj_scale,
                            const int* hand_order, int* nn_idx, float* nn_d2, int* nn_seed,
                            int clip_len, int out_stride, hipStream_t stream);
/* obj_order (Vo) optional, metric-only calls: a permutation of the object vertices, visited in that order (a spatial sort of
 * the rigid mesh makes 64 consecutive vertices a compact patch: scheduling only, the result is the exact minimum) */
int hm_nn_fwd_clips(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
                    float* metric_out, void* workspace, int clip_len, int out_stride, const int* obj_order,
                    hipStream_t stream);
/* The metric-only search on a RIGID object: obj_spheres (B, ceil(Vo/64), 4) = centre + radius, in MESH space, of the groups of
 * 64 vertices taken in `obj_order` (built once by the caller); obj_rot6d (B,3,2) / obj_trans (B,3) / obj_scale (one per clip,
 * used as |s|) = the transform that produced verts_obj (hm_rigid_fwd with abs_scale).  A lane per group carries its sphere
 * into camera space instead of every workgroup reducing all the groups' vertices again.  Scheduling data only: the result is
 * the exact minimum (the spheres only decide which groups are scanned).  All four NULL = hm_nn_fwd_clips.
 * nn_seed (optional; also the last data argument of hm_pair_terms_fwd_clips): (2 + ceil(Vh / 128)) * B ints, zero-filled once by
 * the caller and handed to every call of a loop - the search keeps there, per frame, the vertex pair that held the minimum at the
 * previous call; that pair's distance NOW is an upper bound known before any scan, so that most workgroups scan nothing.  Any
 * content is valid (a pair is a pair); the result does not depend on it. */
int hm_nn_fwd_rigid_clips(const float* verts_hand, const float* verts_obj, int B, int Vh, int Vo, int* nn_idx, float* nn_d2,
                          float* metric_out, void* workspace, int clip_len, int out_stride, const int* obj_order,
                          const float* obj_spheres, const float* obj_rot6d, const float* obj_trans, const float* obj_scale,
                          const int* hand_order, int* nn_seed, hipStream_t stream);
int hm_contact_fwd_clips(const float* verts_hand, const float* verts_obj, const int* nn_idx, int B, int Vh, int Vo,
                         float thresh, float* g_hand, float* g_obj, float* out1, void* workspace, int clip_len,
                         int out_stride, hipStream_t stream);
int hm_collision_fwd_clips(const float* verts0, const int* faces0, int V0, int F0, const float* verts1, const int* faces1,
                           int V1, int F1, int B, float scale_factor, float* g0, float* g1, float* out1, void* workspace,
                           int clip_len, int out_stride, hipStream_t stream);
/* vals (nclips, n+1), log (max_steps, nclips, n+1): per clip the weighted total, then its row of the log */
int hm_log_total_clips(float* vals, const float* weights, int n, const int* step, int max_steps, float* log, int nclips,
                       hipStream_t stream);

/* ------------------------------------------------------------------ measurement / debug hooks (synchronous)
 * hm_bench_sil_kernels: one full silhouette forward + backward to populate the workspace, then `reps` launches of the
 * raster kernel, of the edge-sweep kernel and of the line-source kernel, each series bracketed by two HIP events on
 * `stream`; avg_ms[0..2] (HOST pointer) receive the average launch durations in milliseconds. */
int hm_bench_sil_kernels(const float* verts, const int* faces, const float* K, int B, int V, int F, int S,
                         const float* keep, const float* ref, const float* keep_sum, float* pooled, float* loss_out,
                         const int* work_order, const int* adj_off, const int* adj_items, const int* face_order,
                         const float* upstream, float* grad_verts, void* workspace, int reps, float* avg_ms,
                         hipStream_t stream);
/* hm_debug_sil_timing(1): from then on hm_sil_fwd / hm_sil_bwd record HIP events on their launch stream right before and
 * after k_raster_fwd, k_bwd_lines and k_bwd_sweep, so a caller that issues the optimisation loop launch by launch (not from
 * a captured graph) measures the duration each kernel has inside the loop, next to the work of the other streams;
 * hm_debug_sil_timing_read: durations in ms of the last timed launches {raster, lines, sweep} -> HOST float[3] (waits).
 * Process-wide; hm_debug_sil_timing(0) releases the events. */
/* In-graph timing of k_raster_fwd / k_bwd_lines / k_bwd_sweep: every workgroup (every wave of the persistent sweep kernel)
 * stores the device wall clock (s_memrealtime) at entry and exit into a slot pair of its own inside the silhouette workspace
 * - for launches replayed from a captured hipGraph, where ROCm allows no events.  B, V, F, S as given to
 * hm_sil_workspace_bytes.  hm_sil_timestamps(..., 1, stream) switches on and arms (async memsets; call before every replay),
 * (..., 0, stream) switches off; hm_sil_timestamps_save copies the raw stamps (hm_sil_timestamps_bytes) to `dst` on the
 * device without synchronising; hm_sil_timestamps_read waits for `stream` and returns earliest-start-to-latest-end of the
 * three kernels in microseconds (HOST float[3]) from `saved` or, saved == NULL, from the workspace. */
size_t hm_sil_timestamps_bytes(int B, int V, int F, int S);
int hm_sil_timestamps(void* workspace, int B, int V, int F, int S, int enable, hipStream_t stream);
int hm_sil_timestamps_save(const void* workspace, int B, int V, int F, int S, void* dst, hipStream_t stream);
int hm_sil_timestamps_read(const void* workspace, int B, int V, int F, int S, const void* saved, float* us3, hipStream_t stream);
int hm_debug_sil_timing(int enable);
int hm_debug_sil_timing_read(float* ms3);
int hm_debug_occupancy(int* raster_fwd_blocks, int* sweep_blocks);
int hm_debug_read_partials(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOMAN_AMD_H */
