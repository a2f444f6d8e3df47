"""The hand's gradient chain of one optimisation step in ONE written-out evaluation order (TEST INFRASTRUCTURE -- see
oracle/__init__.py).

`hand_param_grads(model, loss_weights)` evaluates, for an `oracle.model.OracleHOMan` (one hand, optimize_mano, the hand scale a
buffer) at its current parameters, the gradients of
    lw_v2d_hand * loss_v2d_hand + lw_smooth_hand * loss_smooth_hand + lw_inter * loss_inter + lw_pca * loss_pca
- the terms that reach the hand in the step-1 loss sets of reference homan/homan.py:421-508 - with respect to mano_pca_pose,
mano_rot, mano_betas, mano_trans, rotations_hand and translations_hand, as one fixed sequence of IEEE fp32 operations
(oracle/csrc/lbs_exact.c: orc_hand_chain, orc_v2d_unit_grad, orc_inter_rec).  Same mathematics as autograd through
`OracleHOMan.forward` (tests/test_objchain.py: equal within fp32 rounding); every reduction runs in a stated order - the one
csrc/mano.hip and csrc/pair_bodies.h use, so the HIP loop reproduces the values bit for bit.

Chain (file:line of what each stage restates):
  2-D reprojection term, unit gradient per vertex           homan/losses.py:141-164
  temporal smoothness, unit gradient per vertex              homan/lossutils.py:18-36
  coarse interaction: gate + centroid difference per frame    homan/losses.py:199-242 (reaches the rigid pose only: the mesh
                                                              is detached there, homan/homan.py:482-490)
  PCA prior                                                   homan/lossutils.py:39-40
  rigid transform backward, rot6d backward                    homan/utils/camera.py:108-139, utils/geometry.py:9-27
  MANO layer backward (skinning, blend shapes, chain,         homan/manomodel.py:84-151 (smplx-style LBS)
  Rodrigues, PCA)
"""
import ctypes

import numpy as np
import torch

from . import clib
from . import model as o_model
from .objchain import smooth_unit_grad

f32 = np.float32
BLOCK_THREADS = 256          # workgroup size of the interaction term's per-frame reduction (csrc/pair_bodies.h RED_THREADS)


def v2d_unit_grad(verts, camintr, ref2d, image_size):
    N, V = verts.shape[:2]
    out = np.empty((N, V, 3), f32)
    clib.lib().orc_v2d_unit_grad(clib.fptr(verts), clib.fptr(camintr), clib.fptr(ref2d), float(image_size), N, V, clib.fptr(out))
    return out


def inter_records(vh, vo, camintr):
    B = vh.shape[0]
    rec = np.zeros((B, 8), f32)
    clib.lib().orc_inter_rec(clib.fptr(vh), clib.fptr(vo), clib.fptr(camintr), B, vh.shape[1], vo.shape[1],
                             float(o_model.INTERACTION_BBOX_EXPANSION), float(o_model.INTERACTION_Z_THRESH), BLOCK_THREADS,
                             clib.fptr(rec))
    return rec


def hand_param_grads(model, loss_weights, return_stages=False):
    """-> {name: float32 numpy array shaped like the parameter} for the six hand parameters (see the module docstring)."""
    lw = loss_weights
    on = lambda k: lw.get(k, 0.0) > 0
    if (on("lw_collision") or on("lw_contact") or on("lw_depth") or on("lw_sil_hand") or model.hand_nb != 1 or
            not model.optimize_mano or not isinstance(model.mano_betas, torch.nn.Parameter) or
            model.int_scales_hand.requires_grad or model.losses.inter_type != "centroid"):
        raise NotImplementedError("the written-out hand chain covers the step-1 loss sets of a one-hand clip")
    side = model.hand_sides[0]
    c = lambda t: np.ascontiguousarray(t.detach().numpy(), f32)
    with torch.no_grad():
        lbs_verts = model.mano_forward(model.mano_pca_pose, model.mano_rot, model.mano_betas, side)
        mesh_t = lbs_verts + model.mano_trans.unsqueeze(1)
        vh_t, _ = model.get_verts_hand()
        vo_t, _ = model.get_verts_object()
    mesh, vh, vo = c(mesh_t), c(vh_t), c(vo_t)
    B = vh.shape[0]
    K = c(model.camintr)
    terms = []
    if on("lw_smooth_hand") or on("lw_smooth_obj"):
        terms.append((smooth_unit_grad(vh), lw["lw_smooth_hand"]))
    if on("lw_v2d_hand"):
        terms.append((v2d_unit_grad(vh, K, c(model.ref_verts2d_hand), model.image_size), lw["lw_v2d_hand"]))
    rec = inter_records(vh, vo, K) if on("lw_inter") else None
    pca = c(model.mano_pca_pose)
    P = pca.shape[1]
    g_extra = None
    if on("lw_pca"):
        g_extra = np.ascontiguousarray((f32(2.0) * pca) * (f32(1.0) / f32(pca.size)), f32)
    arrs = [np.ascontiguousarray(t, f32) for t, _ in terms]
    ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    ws = np.asarray([w for _, w in terms] or [0.0], f32)
    g_frame = np.ascontiguousarray(rec[:, 2:5]) if rec is not None else None
    out = dict(mano_pca_pose=np.zeros((B, P), f32), mano_rot=np.zeros((B, 3), f32), mano_betas=np.zeros((B, 10), f32),
               mano_trans=np.zeros((B, 3), f32), rotations_hand=np.zeros((B, 6), f32), translations_hand=np.zeros((B, 3), f32))
    lay = model.hands[side]["layout"]
    clib.lib().orc_hand_chain(*[clib.fptr(a) for a in lay[:7]], clib.iptr(lay[7]), clib.fptr(pca), P, clib.fptr(c(model.mano_rot)),
                              clib.fptr(c(model.mano_betas)), clib.fptr(mesh), clib.fptr(c(model.rotations_hand).reshape(B, 6)),
                              float(model.int_scales_hand.detach()[0]), ptrs, clib.fptr(ws), len(arrs),
                              clib.fptr(g_frame) if g_frame is not None else None, 3,
                              float(f32(lw["lw_inter"] / 778)) if rec is not None else 0.0,
                              clib.fptr(g_extra) if g_extra is not None else None, float(lw.get("lw_pca", 0.0)), B,
                              clib.fptr(out["mano_pca_pose"]), clib.fptr(out["mano_rot"]), clib.fptr(out["mano_betas"]),
                              clib.fptr(out["mano_trans"]), clib.fptr(out["rotations_hand"]), clib.fptr(out["translations_hand"]))
    out["rotations_hand"] = out["rotations_hand"].reshape(B, 3, 2)
    out["translations_hand"] = out["translations_hand"].reshape(B, 1, 3)
    if return_stages:
        return out, dict(mesh=mesh, vh=vh, vo=vo, terms=terms, rec=rec)
    return out
