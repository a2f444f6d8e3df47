"""ctypes binding of the C ABI in include/homan_amd.h (libhoman_amd.so).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this
module raises.  Device pointers come from torch tensors (plumbing only: memory + streams).
"""
import ctypes
import os

import torch

from . import build as _build

_LIB = None
_VP, _I, _F, _SZ, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long

# name -> (restype, argtypes)
_SIGNATURES = {
    "hm_sil_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "hm_sil_fwd": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _I, _F, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP,
                        _VP, _I, _I, _VP, _VP]),
    "hm_sil_parts": (_VP, [_VP, _I, _I, _I, _I]),
    "hm_sil_fwd_multi": (_I, [_VP, _I, _I, _VP]),
    "hm_sil_render_bytes": (_SZ, []),
    "hm_sil_hint_near_winding": (_I, [_VP, _I, _VP]),
    "hm_tune_sweep_blocks": (_I, [_I]),
    "hm_tune_raster_lds_pad": (_I, [_I]),
    "hm_tune_raster_reorder": (_I, [_I]),
    "hm_tune_nn_lds_pad": (_I, [_I]),
    "hm_tune_lds_pad": (_I, [_I, _I]),
    "hm_tune_rigid_chunked": (_I, [_I]),
    "hm_debug_sweep_caps": (_I, [_I]),
    "hm_shade_rgb": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _I, _VP, _F, _F, _VP, _VP, _VP, _VP]),
    "hm_rigid_bwd_sil": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _VP]),
    "hm_sil_reduce": (_I, [_I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "hm_depth_bwd": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_depth_bwd_sparse": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_ordinal_depth_bwd_flags": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_ordinal_depth_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "hm_ordinal_depth_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "hm_sil_bwd": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _F, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "hm_bench_sil_kernels": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                  _I, _VP, _VP]),
    "hm_sil_read_boxes": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_debug_occupancy": (_I, [_VP, _VP]),
    "hm_debug_sil_timing": (_I, [_I]),
    "hm_sil_timestamps_bytes": (_SZ, [_I, _I, _I, _I]),
    "hm_sil_timestamps": (_I, [_VP, _I, _I, _I, _I, _I, _VP]),
    "hm_sil_timestamps_save": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_sil_timestamps_read": (_I, [_VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    "hm_debug_sil_timing_read": (_I, [_VP]),
    "hm_debug_read_partials": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_sil_read_idx_map": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_sil_read_faces9": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_sil_read_parts": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hm_sil_invalidate_outputs": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "hm_rigid_fwd": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "hm_rigid_bwd": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _I, _F, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_rigid_workspace_bytes": (_SZ, [_I]),
    "hm_scale_by": (_I, [_VP, _VP, _L, _VP, _VP]),
    "hm_scale2_by": (_I, [_VP, _VP, _VP, _VP, _L, _VP, _VP]),
    "hm_lincomb4": (_I, [_VP, _F, _VP, _F, _VP, _F, _VP, _F, _L, _VP, _VP]),
    "hm_sum_small": (_I, [_VP, _I, _F, _VP, _F, _VP, _VP]),
    "hm_log_total": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP]),
    "hm_mano_fwd": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_mano_state_bytes": (_SZ, [_I]),
    "hm_mano_workspace_bytes": (_SZ, [_I]),
    "hm_mano_bwd": (_I, [_VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_mano_bwd_rigid_clips": (_I, [_VP, _VP, _I, _VP, _VP, _I, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I,
                                     _VP, _VP, _I, _F, _VP, _VP, _I, _VP]),
    "hm_reduce_workspace_bytes": (_SZ, []),
    "hm_v2d_fwd": (_I, [_VP, _VP, _I, _VP, _F, _I, _I, _VP, _VP, _VP, _VP]),
    "hm_smooth_fwd": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "hm_priors_fwd": (_I, [_VP, _L, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_hand_terms_fwd": (_I, [_VP, _VP, _I, _VP, _F, _I, _I, _VP, _VP, _VP, _VP, _VP, _L, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_pose_keep_best": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_pose_keep_best_log": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "hm_offscreen_fwd": (_I, [_VP, _VP, _I, _I, _F, _F, _VP, _VP, _VP]),
    "hm_inter_fwd": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _F, _VP, _VP, _VP, _VP]),
    "hm_inter_bwd": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "hm_nn_fwd": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "hm_contact_fwd": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _VP]),
    "hm_collision_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "hm_collision_fwd": (_I, [_VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _VP]),
    "hm_collision_read_grid": (_I, [_VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "hm_collision_dist_values": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "hm_adam_slot_bytes": (_SZ, []),
    "hm_adam_step": (_I, [_VP, _I, _VP, _F, _F, _F, _I, _I, _VP]),
    "hm_adam_step_log": (_I, [_VP, _I, _VP, _F, _F, _F, _I, _I, _VP, _VP, _I, _I, _VP, _I, _VP]),
    "hm_log_scalars": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    # clip batches (C clips, one launch per kernel): the plain signatures + clip_len [+ out_stride]
    "hm_rigid_fwd_clips": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _I, _VP]),
    "hm_rigid_bwd_clips": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _I, _F, _I, _I, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "hm_rigid_bwd_sil_clips": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP,
                                    _VP, _I, _I, _VP, _F, _VP]),
    "hm_sum_small_clips": (_I, [_VP, _I, _F, _VP, _F, _VP, _I, _VP]),
    "hm_mano_fwd_clips": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "hm_mano_fwd_rows": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "hm_mano_bwd_rows": (_I, [_VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_sil_fwd_clips": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _I, _F, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP,
                              _VP, _VP, _I, _I, _VP, _I, _I, _VP, _VP]),
    "hm_sil_fwd_phase_clips": (_I, [_VP, _VP, _I, _VP, _I, _I, _I, _I, _F, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP,
                                    _VP, _VP, _I, _I, _VP, _I, _I, _VP, _I, _VP]),
    "hm_sil_reduce_clips": (_I, [_I, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_sil_bwd_clips": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _F, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _I, _VP]),
    "hm_sil_bwd_phase_clips": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _F, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _I,
                                    _I, _VP]),
    "hm_v2d_fwd_clips": (_I, [_VP, _VP, _I, _VP, _F, _I, _I, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_smooth_fwd_clips": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_priors_fwd_clips": (_I, [_VP, _L, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_hand_terms_fwd_clips": (_I, [_VP, _VP, _I, _VP, _F, _I, _I, _VP, _VP, _VP, _VP, _VP, _L, _VP, _VP, _VP, _VP, _VP, _VP,
                                     _VP, _VP, _VP, _I, _I, _VP]),
    "hm_pair_terms_fwd_clips": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP,
                                     _VP, _F, _VP, _VP, _VP, _VP, _VP, _L, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                     _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_inter_fwd_clips": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _F, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_nn_fwd_clips": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "hm_nn_fwd_rigid_clips": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "hm_contact_fwd_clips": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_collision_fwd_clips": (_I, [_VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hm_log_total_clips": (_I, [_VP, _VP, _I, _VP, _I, _VP, _I, _VP]),
}


class HomanAmdError(RuntimeError):
    pass


class SilRender(ctypes.Structure):
    """HmSilRender of include/homan_amd.h: one render of hm_sil_fwd_multi (fields as the arguments of hm_sil_fwd_clips)"""
    _fields_ = ([(k, _VP) for k in ("verts", "faces", "K", "keep", "ref", "pooled", "work_order", "pooled_depth", "rigid_rot6d",
                                    "rigid_trans", "rigid_scale", "cam_verts_out", "workspace")]
                + [(k, _I) for k in ("faces_bstride", "B", "V", "F", "S", "mask_shared", "rigid_abs", "persistent_outputs",
                                     "clip_len")]
                + [(k, _F) for k in ("orig_size", "znear", "zfar")])


def sil_renders(renders):
    """[dict of HmSilRender fields (tensors or ints / floats; missing = NULL / 0)] -> ctypes array for hm_sil_fwd_multi.
    The array is read by the library when the call is made (or captured); the tensors must outlive the launches."""
    arr = (SilRender * len(renders))()
    for r, d in zip(arr, renders):
        for k, v in d.items():
            setattr(r, k, ptr(v) if isinstance(v, torch.Tensor) else v)
    assert ctypes.sizeof(SilRender) == lib().hm_sil_render_bytes(), "HmSilRender layout differs from the library's"
    return arr


def lib():
    """Load libhoman_amd.so (after torch, so the HIP runtime already in the process is reused)."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("HOMAN_AMD_LIB", _build.LIB_PATH)      # (A/B runs of kernel variants: tools/ab_build.sh)
        if not os.path.exists(path):
            raise HomanAmdError(
                f"{path} not found: build it with `python -m homan_amd.build` (hipcc, gfx950). "
                "homan_amd has no CPU fallback.")
        _LIB = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(_LIB, name)
            except AttributeError as exc:
                raise HomanAmdError(f"{path} does not export {name}; rebuild it") from exc
            fn.restype, fn.argtypes = res, args
    return _LIB


def exported_symbols():
    return sorted(_SIGNATURES)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "homan_amd ops need contiguous device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def terms(pairs):
    """[(tensor or None, weight), ...] -> (host array of device pointers, host array of floats, n) for the entry points
    that take a weighted list of per-vertex gradients (read by the library at launch time)."""
    live = [(t, w) for t, w in pairs if t is not None]
    n = len(live)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[ptr(t) for t, _ in live])
    ws = (ctypes.c_float * max(n, 1))(*[float(w) for _, w in live])
    return ptrs, ws, n


def check(rc, what):
    if rc != 0:
        raise HomanAmdError(f"{what} failed with code {rc}")


# Graphs are kept alive for the life of the process (HOMAN_KEEP_GRAPHS=0 switches that off).  ROCm 7.0's graph executor has
# crashed in hip::Graph::UpdateStreams at the first replay of a NEW graph after several dozen graphs had been created AND
# destroyed in the process - a test suite, or a fitting process that walks a dataset clip by clip, one stepper per clip.
# Graphs that are never destroyed do not trigger it, and a captured graph here owns no large buffer (the steppers allocate
# before capture): a few kilobytes per fitted clip.
_KEPT_GRAPHS = []


def new_graph():
    g = torch.cuda.CUDAGraph()
    if os.environ.get("HOMAN_KEEP_GRAPHS", "1") != "0":
        _KEPT_GRAPHS.append(g)
    return g


# ... which holds for the FUSED steppers only.  A graph that captures HOMan.forward + autograd (jointopt.GraphStepper, the
# eager-style loop of pose_optimization._graph_loop) owns every activation of the iteration in its private memory pool -
# hundreds of MB per clip / frame - and keeping such graphs alive kept that memory too.  Those captures share ONE pool
# (`torch.cuda.graph(g, pool=autograd_pool())`): the activations are temporaries, freed by the end of the capture, so the next
# capture reuses the same blocks - the graphs stay alive (no destroy, no crash), the memory is bounded by the largest
# iteration plus the few small tensors each capture keeps (tools/soak_dataset.py walks a dataset in graph mode).  Sound
# because steppers are replayed one at a time and nothing but their kept outputs lives across replays.
_AUTOGRAD_POOL = None


def autograd_pool():
    global _AUTOGRAD_POOL
    if _AUTOGRAD_POOL is None:
        _AUTOGRAD_POOL = torch.cuda.graph_pool_handle()
    return _AUTOGRAD_POOL
