"""The object's gradient chain of one optimisation step with order-independent sums (TEST INFRASTRUCTURE -- see
oracle/__init__.py).

`object_pose_grads(model, loss_weights)` evaluates, for an `oracle.model.OracleHOMan` at its current parameters, the
gradients of  lw_sil_obj * loss_sil_obj + lw_smooth_obj * loss_smooth_obj  (the terms that reach the object's pose in the
step-1 loss sets: reference homan/homan.py:421-508 - `loss_inter` sees the object detached, :482-490) with respect to
`rotations_object` and `translations_object`, as ONE fixed sequence of IEEE fp32 operations whose reductions are exact
sums on a power-of-two grid (oracle/csrc/objchain.c).  Same mathematics as autograd through `OracleHOMan.forward`
(tests/test_objchain.py: equal within fp32 rounding); the value no longer depends on the number of host threads, on the
vectorisation of a torch reduction or on libm, and the HIP kernels reproduce it bit for bit.

Chain (file:line of what each stage restates):
  verts = (|s| v) R(rot6d) + t                       homan/homan.py:298-307, utils/camera.py:108-139   (oracle.model, bit-exact)
  NDC faces, fill_back, face-index map                 neural_renderer projection + rasteriser            (oracle.nmr, nmr_raster.c)
  pooled silhouette, dL/dsilhouette                    homan/losses.py:183-197: image = keep * rend;
                                                       loss = sum((image - ref)^2) / sum(keep) / B
  per-sample gradient, edge-sweep pseudo-gradient      neural_renderer backward                          (objchain.c)
  vertex gather, projection backward, + smoothness     homan/lossutils.py:18-36
  rigid backward, rot6d backward                       utils/geometry.py:9-27                            (objchain.c)
"""
import ctypes

import numpy as np
import torch

from . import clib
from . import nmr as o_nmr

f32 = np.float32


def build_adjacency(faces, num_verts):
    """CSR vertex -> (face * 3 + corner), items ascending (the order does not matter to the exact sums)."""
    flat = np.asarray(faces, np.int64).reshape(-1)
    order = np.argsort(flat, kind="stable")
    off = np.zeros(num_verts + 1, np.int32)
    np.cumsum(np.bincount(flat, minlength=num_verts), out=off[1:])
    return off, order.astype(np.int32)


def smooth_unit_grad(verts):
    """d mean((v[t+1] - v[t])^2) / d v, element-wise (reference homan/lossutils.py:18-36): 2 * g / count with
    g = -(v[t+1] - v[t]) + (v[t] - v[t-1]), count = (B - 1) * V * 3."""
    v = np.ascontiguousarray(verts, f32)
    B = v.shape[0]
    cnt = (B - 1) * v.shape[1] * 3
    inv_cnt = f32(1.0) / f32(cnt) if cnt > 0 else f32(0.0)
    g = np.zeros_like(v)
    if B > 1:
        g[:-1] = f32(0.0) - (v[1:] - v[:-1])
        g[1:] = g[1:] + (v[1:] - v[:-1])
    return (f32(2.0) * g) * inv_cnt


def sample_gradient(idx, keep, ref, upstream, keep_sum, B):
    """dL/dalpha on the (B, 2S, 2S) sample grid of the rasteriser (before its vertical flip) for
    L = upstream * sum((keep * pool(alpha) - ref)^2) / keep_sum / B:  per pixel g = ((upstream * 2) * dimg) / keep_sum / B with
    dimg = keep * (keep * pool - ref), per sample 0.25 * g (2x2 average pool), the operations in this order."""
    alpha = (idx >= 0).astype(f32)[:, ::-1]                                   # vertical flip
    n = alpha.shape[1] // 2
    a4 = alpha.reshape(idx.shape[0], n, 2, n, 2)
    pool = f32(0.25) * (((a4[:, :, 0, :, 0] + a4[:, :, 0, :, 1]) + a4[:, :, 1, :, 0]) + a4[:, :, 1, :, 1])   # exact: quarters
    dimg = keep * (keep * pool - ref)
    g = ((f32(upstream) * f32(2.0)) * dimg) / f32(keep_sum) / f32(B)
    gs = f32(0.25) * g
    full = np.repeat(np.repeat(gs, 2, axis=1), 2, axis=2)[:, ::-1]            # back onto the unflipped sample grid
    return np.ascontiguousarray(full, f32), pool


def pseudo_gradient_exact(faces_ndc, idx, grad_alpha, num_faces, eps=o_nmr.DEFAULT_EPS, log2q=0):
    """(B, 2F, 9) NDC faces (mesh faces, then their reversed copies), index map, per-sample gradient -> (B, F, 3, 2) float64."""
    B, S2 = idx.shape[0], idx.shape[1]
    parts = np.zeros((B, num_faces, 3, 2), np.float64)
    clib.lib().orc_nmr_grad_faces_alpha_exact(clib.fptr(faces_ndc), clib.iptr(idx), clib.fptr(grad_alpha), B, num_faces, S2,
                                              eps, log2q, clib.dptr(parts))
    return parts


def rigid_bwd_sil_exact(mesh, rot6d, scale, abs_scale, terms, parts, adj, cam_verts, K, orig_size, num_faces, log2q=0):
    """terms: [(array (B,V,3) f32, weight)] -> g_rot6d (B,6), g_trans (B,3), g_scale_part (B), g_verts (B,V,3)"""
    B, V = mesh.shape[:2]
    arrs = [np.ascontiguousarray(t, f32) for t, _ in terms]
    ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    ws = np.asarray([w for _, w in terms] or [0.0], f32)
    g_rot, g_tr, g_sc, g_v = np.zeros((B, 6), f32), np.zeros((B, 3), f32), np.zeros(B, f32), np.zeros((B, V, 3), f32)
    clib.lib().orc_rigid_bwd_sil_exact(clib.fptr(mesh), clib.fptr(rot6d), float(scale), int(abs_scale), ptrs, clib.fptr(ws),
                                       len(arrs), clib.dptr(parts) if parts is not None else None, clib.iptr(adj[0]),
                                       clib.iptr(adj[1]), clib.fptr(cam_verts), clib.fptr(K), float(orig_size), num_faces, B, V,
                                       log2q, clib.fptr(g_rot), clib.fptr(g_tr), clib.fptr(g_sc), clib.fptr(g_v))
    return g_rot, g_tr, g_sc, g_v


def object_pose_grads(model, loss_weights, log2q=0, return_stages=False, contact_obj=None, inter_rec=None, depth_obj=None,
                      obj_terms=None):
    """-> {"rotations_object": (B,3,2), "translations_object": (B,1,3)[, "int_scales_object": (1,)]} float32 numpy (see the module
    docstring).  contact_obj: d loss_contact / d object vertices (B,V,3) of the step-2 sets (oracle/handchain.py pair_terms).
    With a free object scale (optimize_object_scale; the object is then NOT detached in the interaction term, homan/homan.py:
    482-490) inter_rec (B,8) are that term's per-frame records (oracle/handchain.py inter_records)."""
    lw = loss_weights
    free_scale = bool(model.optimize_object_scale)
    if obj_terms is None and ((lw.get("lw_depth", 0) > 0 and depth_obj is None) or
                              (lw.get("lw_contact", 0) > 0 and contact_obj is None) or
                              (free_scale and lw.get("lw_inter", 0) > 0 and inter_rec is None)):
        raise NotImplementedError("the written-out object chain covers silhouette + smoothness (+ the contact / interaction "
                                  "terms' gradients on the object's vertices, handed in)")
    with torch.no_grad():
        verts_t, _ = model.get_verts_object()
        rend = model.losses.renderer
        faces_t = rend._ndc_faces(verts_t, model.faces_object, model.camintr_rois_object, None, None, None, None)
    verts = np.ascontiguousarray(verts_t.numpy(), f32)
    B, V = verts.shape[:2]
    F = model.faces_object.shape[1]
    faces_ndc = np.ascontiguousarray(faces_t.numpy().reshape(B, 2 * F, 9), f32)
    S = rend.image_size
    idx = np.empty((B, 2 * S, 2 * S), np.int32)
    dep = np.empty((B, 2 * S, 2 * S), f32)
    clib.lib().orc_nmr_face_index_map(clib.fptr(faces_ndc), B, 2 * F, 2 * S, rend.near, rend.far, clib.iptr(idx), clib.fptr(dep))
    stages = dict(verts=verts, faces_ndc=faces_ndc, idx=idx)
    parts = None
    if lw.get("lw_sil_obj", 0) > 0:
        keep = np.ascontiguousarray(model.keep_mask_object.numpy(), f32)
        ref = np.ascontiguousarray(model.ref_mask_object.numpy(), f32)
        keep_sum = f32(model.keep_mask_object.sum().item())        # (a count: exact in fp32 below 2^24 pixels)
        ga, pool = sample_gradient(idx, keep, ref, lw["lw_sil_obj"], keep_sum, B)
        parts = pseudo_gradient_exact(faces_ndc, idx, ga, F, rend.rasterizer_eps, log2q)
        stages.update(grad_alpha=ga, pooled=pool, parts=parts)
    terms = []
    if lw.get("lw_smooth_obj", 0) > 0 or lw.get("lw_smooth_hand", 0) > 0:
        terms.append((smooth_unit_grad(verts), lw["lw_smooth_obj"]))
    if obj_terms is not None:           # (two hands: the pair terms' gradients on the object, ready-made, oracle/handchain.py)
        terms.extend((np.ascontiguousarray(a, f32), w) for a, w in obj_terms)
    elif lw.get("lw_contact", 0) > 0:
        terms.append((np.ascontiguousarray(contact_obj, f32), lw["lw_contact"]))
    if obj_terms is None and free_scale and lw.get("lw_inter", 0) > 0 and model.losses.inter_type != "centroid":
        # inter_type "min" (homan/losses.py:219-221) with a free scale: the closest pair's pull reaches the object too - minus the
        # hand's pull, on the ONE object vertex j* of every gated frame (oracle/handchain.py min_pair_pull names the pair; the fused
        # loop scatters the same vector, homan_amd/fused.py)
        from .handchain import min_pair_pull
        with torch.no_grad():
            vh = np.ascontiguousarray(model.get_verts_hand()[1].numpy(), f32)          # (the mesh-detached twin: same floats)
        _, st = min_pair_pull(vh, verts, inter_rec, lw["lw_inter"])
        go = np.zeros((B, V, 3), f32)
        go[np.arange(B), st["j_star"]] = f32(0.0) - st["pull"]
        terms.append((go, 1.0))
    elif obj_terms is None and free_scale and lw.get("lw_inter", 0) > 0:
        # d (lw_inter * loss_inter) / d object vertex = -lw_inter * gate * 2 (c_hand - c_obj) / 3 / V, the same for every vertex
        gi = (f32(0.0) - f32(lw["lw_inter"])) * np.ascontiguousarray(inter_rec[:, 2:5], f32) / f32(V)
        terms.append((np.ascontiguousarray(np.broadcast_to(gi[:, None, :], (B, V, 3)), f32), 1.0))
    if lw.get("lw_depth", 0) > 0 and obj_terms is None:   # d (lw_depth * loss_depth) / d object vertices (oracle/depthchain.py), times its weight (two hands: among obj_terms)
        terms.append((np.ascontiguousarray(depth_obj, f32), 1.0))
    adj = build_adjacency(model.faces_object[0].numpy(), V)
    mesh = np.ascontiguousarray(model.verts_object_og.detach().numpy(), f32)
    rot6d = np.ascontiguousarray(model.rotations_object.detach().numpy().reshape(B, 6), f32)
    K = np.ascontiguousarray(model.camintr_rois_object.numpy(), f32)
    g_rot, g_tr, g_sc, g_v = rigid_bwd_sil_exact(mesh, rot6d, float(model.int_scales_object.detach()[0]), 1, terms, parts, adj,
                                                 verts, K, 1.0, F, log2q)
    out = {"rotations_object": g_rot.reshape(B, 3, 2), "translations_object": g_tr.reshape(B, 1, 3)}
    if free_scale:
        # the clip's scale gradient: the frames' partial sums (one 64-thread block sum, csrc/geometry.hip k_sum_small) + the prior
        # of homan/lossutils.py:107-109 (d (s - mean)^2 = 2 (s - mean))
        g = f32(1.0) * f32(clib.lib().orc_block_sum(clib.fptr(np.ascontiguousarray(g_sc, f32)), B, 64))
        if lw.get("lw_scale_obj", 0) > 0:
            d0 = f32(model.int_scales_object.detach().numpy().reshape(-1)[0]) - f32(np.asarray(model.int_scale_object_mean).reshape(-1)[0])
            g = g + f32(lw["lw_scale_obj"]) * (f32(2.0) * d0)
        out["int_scales_object"] = np.asarray([g], f32)
    if return_stages:
        stages.update(g_verts=g_v, terms=terms)
        return out, stages
    return out
