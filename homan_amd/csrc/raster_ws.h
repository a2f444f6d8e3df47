// raster_ws.h -- host side of the rasteriser's translation units: the layout of the caller's workspace (hm_sil_workspace_bytes),
// the launch hints, and the launchers every unit exports to raster_api.hip (plain host functions: a unit owns its kernels).
#pragma once
#include "raster_common.h"

#define HM_INTERNAL __attribute__((visibility("hidden")))      // shared between the units, not part of the C ABI

// persistent sweep waves: 4 per SIMD.  More does not speed the sweep up and starves the concurrent hand-side kernels
// of wave slots (they run on a second stream of the same hipGraph).
#ifndef SWEEP_BLOCKS
#define SWEEP_BLOCKS 1280
#endif
// Launch hints (hm_tune_*, raster_api.hip): scheduling only, no effect on results; per calling thread, read when an entry point
// is called (or captured).
struct RasterTune {
    int raster_lds_pad;     // bytes of unused dynamic LDS per k_raster_fwd workgroup (hm_tune_raster_lds_pad)
    int raster_reorder;     // adaptive launch order of the forward raster (hm_tune_raster_reorder)
    int sweep_blocks;       // persistent workgroups of k_bwd_sweep (hm_tune_sweep_blocks)
};
HM_INTERNAL RasterTune& hm_raster_tune();
// test hook (hm_debug_sweep_caps): > 0 shrinks the unit table and the partial-slot table of the sweep work list
HM_INTERNAL int& hm_sweep_cap_override();
// timestamp slots of hm_sil_timestamps: {start, end} per raster workgroup, per lines workgroup, per sweep wave
#define TS_SWEEP_WGS 4096
static inline size_t ts_raster_units(int B, int S) { return (size_t)B * (S / 8) * (S / 8) / RASTER_WAVES; }
static inline size_t ts_lines_units(int B, int F, int S) { return (size_t)B * F / 256 + 2 * (size_t)B + 64 + (size_t)B * S / 2 + 8; }
static inline size_t ts_units(int B, int F, int S) { return ts_raster_units(B, S) + ts_lines_units(B, F, S) + 4 * TS_SWEEP_WGS; }
static inline size_t sweep_ucap(int B, int F) { return (size_t)B * F * 4 + 1024; }
static inline size_t sweep_slot_cap(int B, int F) { return sweep_ucap(B, F) + (size_t)B * F; }

// workspace layout helper (bytes), all chunks 256-byte aligned
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SilWs {
    unsigned int* counter; float* frame_rec;
    float* ndc; float* faces9; FaceBox* boxes; int* idx_map; unsigned short* alpha16; float* dimg;
    float* partials; float* gimg; unsigned short* planes; double* parts;
    unsigned char* owned; int* bin_cnt; unsigned int* bin_done; unsigned char* region_state; int* bin_list;
    uint4* lrec; SweepSrc* srcs; unsigned short* lsum;
    SweepList sweep;
    unsigned long long* ts;      // {start, end} slots: raster workgroups | lines workgroups | sweep waves
    int* wo_dyn; int* wo_tmp; unsigned int* wg_cost;      // adaptive raster launch order (hm_tune_raster_reorder)
};
static inline SilWs carve(void* ws, int B, int V, int F, int S)
{
    const size_t is = 2 * (size_t)S;
    char* p = (char*)ws;
    SilWs w;
    w.counter = (unsigned int*)p; p += 256;
    w.frame_rec = (float*)p; p += al256((size_t)B * 16);
    w.ndc = (float*)p; p += al256((size_t)B * V * 3 * 4);
    w.faces9 = (float*)p; p += al256((size_t)B * F * 9 * 4);
    w.boxes = (FaceBox*)p; p += al256((size_t)B * F * 8);
    w.idx_map = (int*)p; p += al256((size_t)B * is * is * 4);
    w.alpha16 = (unsigned short*)p; p += al256((size_t)B * is * (is / 16) * 2);
    w.dimg = (float*)p; p += al256((size_t)B * S * S * 4);
    w.partials = (float*)p; p += al256((size_t)B * (S / 8) * (S / 8) * 16);
    w.gimg = (float*)p; p += al256((size_t)B * is * is * 4);
    w.planes = (unsigned short*)p; p += al256((size_t)B * is * (is / 16) * 4) * 2;      // (B, T, T, 4, 16) u16
    w.parts = (double*)p; p += al256((size_t)B * F * 24 * 4);      // (6 doubles per face for the sweeps; 9 floats for the depth backward)
    w.owned = (unsigned char*)p; p += al256((size_t)B * F * 2);
    w.bin_cnt = (int*)p; w.bin_done = (unsigned int*)(p + (size_t)B * SR_MAX * 4); p += al256((size_t)B * SR_MAX * 8);
    w.region_state = (unsigned char*)p; p += al256((size_t)B * (S / 16) * (S / 16));
    w.bin_list = (int*)p; p += al256((size_t)B * SR_MAX * F * 4);
    w.lrec = (uint4*)p; p += al256(4 * (size_t)B * is * (is / 64) * 16);
    w.lsum = (unsigned short*)p; p += al256((size_t)B * 2 * is * 16);
    w.srcs = (SweepSrc*)p; p += al256(4 * (size_t)B * is * is * sizeof(SweepSrc));
    w.sweep.tab = (SweepFace*)p; p += al256((size_t)B * F * sizeof(SweepFace));
    w.sweep.offs = (int*)p; p += al256((size_t)B * F * 4);
    w.sweep.tickets = (unsigned int*)p; p += al256((size_t)B * F * 4);
    w.sweep.ufirst = (unsigned int*)p; p += al256(sweep_ucap(B, F) * 4);
    w.sweep.upart = (float*)p; p += al256(sweep_slot_cap(B, F) * 24);
    w.ts = (unsigned long long*)p; p += al256(ts_units(B, F, S) * 16);
    w.wo_dyn = (int*)p; p += al256(ts_raster_units(B, S) * 4);
    w.wo_tmp = (int*)p; p += al256(ts_raster_units(B, S) * 4);
    w.wg_cost = (unsigned int*)p;
    w.sweep.cnt = (unsigned long long*)(w.counter + 16);        // zero between launches (re-armed by the last compaction block)
    w.sweep.done = w.counter + 18;
    w.sweep.total = (unsigned long long*)(w.counter + 20);
    w.sweep.ucap = (int)sweep_ucap(B, F);
    w.sweep.slot_cap = (int)sweep_slot_cap(B, F);
    if (hm_sweep_cap_override() > 0) {          // test hook (hm_debug_sweep_caps): force the beyond-capacity paths
        w.sweep.ucap = min(w.sweep.ucap, hm_sweep_cap_override());
        w.sweep.slot_cap = min(w.sweep.slot_cap, hm_sweep_cap_override());
    }
    return w;
}

// 32-bit byte offsets (the W32 instantiations of the line expansion and the sweeps) while the largest array they index - the
// per-line source arrays, 4 B is^2 records of 12 bytes - stays below 4 GB.  HOMAN_FORCE_W64=1 (read once; a test hook)
// takes the 64-bit instantiations regardless: tests/test_raster_gpu.py runs the bit-exactness tests through both.
static inline bool hm_offsets_fit_32(int B, int S)
{
    static const bool force64 = [] { const char* e = getenv("HOMAN_FORCE_W64"); return e && atoi(e) != 0; }();
    return !force64 && 4.0 * B * (2.0 * S) * (2.0 * S) * sizeof(SweepSrc) < 4.0e9;
}

// ---------------------------------------------------------------- launchers (one per kernel family, defined next to the kernels)
// raster_setup.hip: k_setup_faces, grid (nfb + vertex blocks, B).  bins == NULL: no super-region bins (is > 1024)
HM_INTERNAL void hm_launch_setup_faces(const SilWs& w, const float* verts, const float* K, float orig_size, const int* faces,
                                       int faces_bstride, int B, int V, int F, int is, int* bins, const float* rigid_rot6d,
                                       const float* rigid_trans, const float* rigid_scale, int rigid_abs, int clip_len,
                                       float* cam_verts_out, hipStream_t stream);
#define HM_MAX_RENDERS 4      // renders of one hm_sil_fwd_multi launch pair
// one render of hm_sil_fwd_multi: the layout include/homan_amd.h declares (hm_sil_render_bytes() lets a binding check its own)
struct HmSilRender {
    const float* verts; const int* faces; const float* K; const float* keep; const float* ref; float* pooled;
    const int* work_order; float* pooled_depth; const float* rigid_rot6d; const float* rigid_trans; const float* rigid_scale;
    float* cam_verts_out; void* workspace;
    int faces_bstride, B, V, F, S, mask_shared, rigid_abs, persistent_outputs, clip_len;
    float orig_size, znear, zfar;
};
struct SetupFacesArgs {
    const float* verts; const float* K; float orig_size; const int* faces; int faces_bstride; int B, V, F, is; int* bins;
    const float* rigid_rot6d; const float* rigid_trans; const float* rigid_scale; int rigid_abs; int clip_len; float* cam_verts_out;
};
struct SetupFacesK {         // k_setup_faces' own argument block (one render)
    const float* verts; const float* K; float orig_size; const int* faces; int faces_bstride; int B, V, F, is; float* faces9;
    FaceBox* boxes; unsigned char* owned; int* bin_cnt; int* bin_list; const float* rigid_rot6d; const float* rigid_trans;
    const float* rigid_scale; int rigid_abs; int clip_len; float* cam_out; int nfb;
};
struct SetupFacesMulti { SetupFacesK r[HM_MAX_RENDERS]; int first[HM_MAX_RENDERS + 1]; int nblk[HM_MAX_RENDERS]; int n; };
HM_INTERNAL void hm_launch_setup_faces_multi(const SilWs* w, const SetupFacesArgs* a, int n, hipStream_t stream);
// raster_fwd.hip: k_raster_fwd over all (frame, region) pairs; fused = masked-MSE / IoU partials from keep / ref
struct RasterFwdArgs {
    int B, F, S;
    float znear, zfar;
    float* pooled; const float* keep; const float* ref; bool fused;
    const int* work_order; float* pooled_depth; int* bins; int reset_bins; int persistent; float* alpha_full; int mask_shared;
    bool per_sample_grad;       // fused per-sample L2 (rendering without anti-aliasing): dimg_full = the workspace's gimg
    bool reorder;               // record workgroup costs for the adaptive launch order
    int lds_pad;
};
HM_INTERNAL void hm_launch_raster_fwd(const SilWs& w, const RasterFwdArgs& a, hipStream_t stream);
// the kernel's own argument block (one render), and several of them for k_raster_fwd_multi (hm_sil_fwd_multi): the workgroups of
// render g are [first[g], first[g + 1])
struct RasterFwdK {
    const float* faces9; const FaceBox* boxes; int B, F, S; float znear, zfar; int* idx_map; unsigned short* alpha16; float* pooled;
    const float* keep; const float* ref; float* dimg; float* partials; const int* work_order; unsigned char* owned;
    float* pooled_depth; unsigned short* planes; int* bin_cnt; const int* bin_list; unsigned int* done; int reset_bins;
    unsigned char* region_state; int persistent; float* alpha_full; int mask_shared; float* dimg_full; const unsigned int* hint;
    unsigned long long* ts_slots; int* wo_dyn; unsigned int* wg_cost;
};
struct RasterFwdMulti { RasterFwdK r[HM_MAX_RENDERS]; int first[HM_MAX_RENDERS + 1]; int n; };
HM_INTERNAL void hm_launch_raster_fwd_multi(const SilWs* w, const RasterFwdArgs* a, int n, hipStream_t stream);
HM_INTERNAL void hm_launch_sil_reduce(const SilWs& w, int B, int S, const float* keep_sum, float* loss_out, float* frame_out,
                                      int clip_len, int out_stride, hipStream_t stream);
HM_INTERNAL void hm_launch_shade_rgb(const SilWs& w, const float* verts, const int* faces, int faces_bstride, const float* textures,
                                     int B, int V, int F, int S, const float* light_dir, float amb, float dirw,
                                     const float* background, float* rgb, hipStream_t stream);
// raster_lines.hip: sample-gradient masks of the generic backward; line expansion + work list (+ the loss reduction in front)
HM_INTERNAL void hm_launch_bwd_masks(const SilWs& w, const float* gin, int mode, const float* upstream, const float* keep_sum,
                                     int B, int S, int clip_len, hipStream_t stream);
HM_INTERNAL void hm_launch_lines(const SilWs& w, int B, int F, int S, int mode, const float* upstream, const float* keep_sum,
                                 int clip_len, hipStream_t stream, float* loss_out = nullptr, int out_stride = 0);
// raster_sweep.hip: edge sweeps over the work list; vertex gather + projection backward
HM_INTERNAL void hm_launch_sweep(const SilWs& w, int B, int F, int S, float eps, int sum_log2q, hipStream_t stream);
HM_INTERNAL void hm_launch_bwd_gather(const SilWs& w, const int* adj_off, const int* adj_items, const float* verts, const float* K,
                                      int B, int V, int F, float orig_size, float* grad_ndc, float* grad_verts, hipStream_t stream);
HM_INTERNAL int hm_sweep_occupancy(int* blocks_per_cu);
HM_INTERNAL int hm_raster_fwd_occupancy(int* blocks_per_cu);
