"""The hand's gradient chain of one optimisation step in ONE written-out evaluation order (TEST INFRASTRUCTURE -- see
oracle/__init__.py).

`hand_param_grads(model, loss_weights)` evaluates, for an `oracle.model.OracleHOMan` (one hand, optimize_mano, the hand scale a
buffer) at its current parameters, the gradients of
    lw_v2d_hand * loss_v2d_hand + lw_smooth_hand * loss_smooth_hand + lw_inter * loss_inter + lw_pca * loss_pca
    [+ lw_collision * loss_collision + lw_contact * loss_contact]
- the terms that reach the hand in the step-1 [step-2] loss sets of reference homan/homan.py:421-508 - with respect to mano_pca_pose,
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
  collision: trilinear samples of the object's clamped SDF     homan/lossutils.py:43-64, interactions/scenesdf.py:77-148
  contact: nearest object vertex, tanh-saturated distance      homan/lossutils.py:112-130, interactions/contactloss.py:60-79,
                                                              162-163, 228-257 (also reaches the object: oracle/objchain.py)
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


def v2d_unit_grad(verts, camintr, ref2d, image_size, hand_nb=1):
    N, V = verts.shape[:2]
    out = np.empty((N, V, 3), f32)
    clib.lib().orc_v2d_unit_grad(clib.fptr(verts), clib.fptr(camintr), clib.fptr(ref2d), float(image_size), N, V, hand_nb,
                                 clib.fptr(out))
    return out


def smooth_unit_grad_rows(verts, hand_nb):
    """temporal smoothness of rows interleaved frame-major (hand i of every frame = rows i, i + hand_nb, ...): the neighbours of
    a row are hand_nb rows away (csrc/pair_bodies.h smooth_body); hand_nb = 1 is oracle.objchain.smooth_unit_grad"""
    v = np.ascontiguousarray(verts, f32)
    N, h = v.shape[0], hand_nb
    cnt = (N - h) * v.shape[1] * 3
    inv_cnt = f32(1.0) / f32(cnt) if cnt > 0 else f32(0.0)
    g = np.zeros_like(v)
    if N > h:
        g[:-h] = f32(0.0) - (v[h:] - v[:-h])
        g[h:] = g[h:] + (v[h:] - v[:-h])
    return (f32(2.0) * g) * inv_cnt


def inter_records(vh, vo, camintr):
    B = vh.shape[0]
    rec = np.zeros((B, 8), f32)
    clib.lib().orc_inter_rec(clib.fptr(vh), clib.fptr(vo), clib.fptr(camintr), B, vh.shape[1], vo.shape[1],
                             float(o_model.INTERACTION_BBOX_EXPANSION), float(o_model.INTERACTION_Z_THRESH), BLOCK_THREADS,
                             clib.fptr(rec))
    return rec


COLLISION_THRESH = 0.020     # reference homan/interactions/contactloss.py compute_contact_loss default (lossutils.py:112-130)
SDF_SCALE_FACTOR = 0.2       # reference homan/interactions/scenesdf.py:77 SDFSceneLoss.forward(scale_factor=0.2)
SDF_N = 32


def nearest_object_vertex(vh, vo):
    B, Vh = vh.shape[:2]
    idx = np.empty((B, Vh), np.int32)
    clib.lib().orc_nn_search(clib.fptr(vh), clib.fptr(vo), B, Vh, vo.shape[1], clib.iptr(idx))
    return idx


def contact_unit_grads(vh, vo, idx):
    """-> (d loss_contact / d hand vertices (B,Vh,3), d loss_contact / d object vertices (B,Vo,3))"""
    B, Vh, Vo = vh.shape[0], vh.shape[1], vo.shape[1]
    gh, go = np.empty((B, Vh, 3), f32), np.empty((B, Vo, 3), f32)
    clib.lib().orc_contact_grads(clib.fptr(vh), clib.fptr(vo), clib.iptr(idx), B, Vh, Vo, COLLISION_THRESH, B, clib.fptr(gh),
                                 clib.fptr(go))
    return gh, go


def scene_box(verts):
    """(B,V,3) -> (B,4) = centre xyz, scale of the normalised box (scenesdf.py:100-112; csrc/sdf.hip k_sdf_boxes)"""
    lo, hi = verts.min(1), verts.max(1)
    ctr = (lo + hi) / f32(2.0)
    sc = ((hi - lo) * ((f32(1.0) + f32(SDF_SCALE_FACTOR)) * f32(0.5))).max(1)
    return np.ascontiguousarray(np.concatenate([ctr, sc[:, None]], 1), f32)


def collision_unit_grad(points, mesh_verts, mesh_faces):
    """d loss_collision / d points (B,V,3): `points` sampled in the clamped SDF of the mesh (mesh_verts (B,Vm,3), mesh_faces (F,3));
    the other ordered pair of the scene reaches the mesh that is detached in the step-2 objective (homan/homan.py:476-480)."""
    from . import sdfgrid
    box = scene_box(mesh_verts)
    local = np.ascontiguousarray((mesh_verts - box[:, None, :3]) / box[:, None, 3:4], f32)
    phi = sdfgrid.SDF(clamp_outside=True)(torch.as_tensor(mesh_faces).int(), torch.from_numpy(local), SDF_N).numpy()
    phi = np.ascontiguousarray(np.maximum(phi, f32(0.0)), f32)
    g = np.empty_like(points)
    clib.lib().orc_sdf_sample_grad(clib.fptr(points), clib.fptr(box), clib.fptr(phi), points.shape[0], points.shape[1], SDF_N,
                                   clib.fptr(g))
    return g, dict(box=box, phi=phi)


def min_pair_pull(vh, vo, rec, lw_inter):
    """inter_type "min": -> (g_rigid (B,Vh,3): lw_inter * d (gate * |h_i* - o_j*|^2) / d h on the one vertex i* per frame, zeros
    elsewhere; dict(i_star, j_star, pull)).  (i*, j*) = the closest pair: j by the nearest-vertex search of every hand vertex,
    i = the first hand vertex with the smallest of those squared distances."""
    B, Vh = vh.shape[:2]
    idx = np.empty((B, Vh), np.int32)
    d2 = np.empty((B, Vh), f32)
    clib.lib().orc_nn_search_d2(clib.fptr(vh), clib.fptr(vo), B, Vh, vo.shape[1], clib.iptr(idx), clib.fptr(d2))
    rows = np.arange(B)
    i_star = d2.argmin(1)
    j_star = idx[rows, i_star]
    flags = (rec[:, 0] != 0).astype(f32)
    diff = vh[rows, i_star] - vo[rows, j_star]
    pull = (f32(2.0 * lw_inter) * diff) * flags[:, None]
    g = np.zeros((B, Vh, 3), f32)
    g[rows, i_star] = pull
    return g, dict(i_star=i_star, j_star=j_star, pull=pull)


def pair_terms(model, vh, vo, loss_weights):
    """The step-2 terms between the two meshes at their current vertices -> dict(col_hand, con_hand, con_obj, nn_idx, ...)"""
    out = {}
    if loss_weights.get("lw_collision", 0.0) > 0:
        out["col_hand"], out["col_stage"] = collision_unit_grad(vh, vo, model.faces_object[0].numpy())
    if loss_weights.get("lw_contact", 0.0) > 0:
        out["nn_idx"] = nearest_object_vertex(vh, vo)
        out["con_hand"], out["con_obj"] = contact_unit_grads(vh, vo, out["nn_idx"])
    return out


def two_hand_terms(model, loss_weights):
    """Two hands per frame (rows interleaved frame-major, right / left through their own models; reference homan/homan.py:62-63,
    341-358; lossutils.py:53-59, 116-127; losses.py:199-242): everything between the meshes, per hand, in the order of the fused
    loop's two-hand path.  -> dict(vh, vo, mesh, terms [(array (N,778,3), weight)], rec (N,8), obj_terms [(array (B,Vo,3), weight)])"""
    lw = loss_weights
    on = lambda k: lw.get(k, 0.0) > 0
    h = 2
    c = lambda t: np.ascontiguousarray(t.detach().numpy(), f32)
    with torch.no_grad():
        per_hand = [model.mano_forward(model.mano_pca_pose[i::h], model.mano_rot[i::h], model.mano_betas[i::h], side)
                    for i, side in enumerate(model.hand_sides)]
        mesh_t = torch.stack(per_hand).transpose(0, 1).contiguous().view(-1, 778, 3) + model.mano_trans.unsqueeze(1)
        vh_t, _ = model.get_verts_hand()
        vo_t, _ = model.get_verts_object()
    mesh, vh, vo = c(mesh_t), c(vh_t), c(vo_t)
    N, B, Vo = vh.shape[0], vo.shape[0], vo.shape[1]
    K = c(model.camintr)
    hands = [np.ascontiguousarray(vh[i::h]) for i in range(h)]
    terms, obj_terms = [], []
    if on("lw_smooth_hand") or on("lw_smooth_obj"):
        terms.append((smooth_unit_grad_rows(vh, h), lw["lw_smooth_hand"]))
    if on("lw_v2d_hand"):
        terms.append((v2d_unit_grad(vh, K, c(model.ref_verts2d_hand), model.image_size, h), lw["lw_v2d_hand"]))
    stages = {}
    if on("lw_collision"):
        rev = np.ascontiguousarray(np.asarray(model.closed_faces)[:, ::-1])          # both hands in reversed winding (:53-59)
        fo = model.faces_object[0].numpy()
        d = [collision_unit_grad(hands[0], hands[1], rev)[0], collision_unit_grad(hands[1], hands[0], rev)[0],
             collision_unit_grad(hands[0], vo, fo)[0], collision_unit_grad(hands[1], vo, fo)[0]]
        cols = []
        for a, b in ((d[0], d[1]), (d[2], d[3])):          # from the hand-hand scene, then from each hand's scene with the object
            u = np.empty((N, 778, 3), f32)
            u[0::h], u[1::h] = a, b
            cols.append(u)
        if on("lw_depth"):      # (the hands' rigid backward sums five terms: with the depth term the two collision buffers share a slot)
            terms.append((cols[0] + cols[1], lw["lw_collision"]))
        else:
            terms.extend((u, lw["lw_collision"]) for u in cols)
        stages["col"] = d
    dep = None
    if on("lw_depth"):          # the ordinal depth term over the three layers (oracle/depthchain.py), already times its weight
        from . import depthchain
        dep = depthchain.depth_vertex_grads_layers(model, lw["lw_depth"])
        u = np.empty((N, 778, 3), f32)
        u[0::h], u[1::h] = dep[1], dep[2]
        terms.append((u, 1.0))
        stages["depth"] = dep
    if on("lw_contact"):
        u = np.empty((N, 778, 3), f32)
        stages["nn_idx"], stages["con_obj"] = [], []
        for i in range(h):
            idx = nearest_object_vertex(hands[i], vo)
            gh, go = contact_unit_grads(hands[i], vo, idx)
            u[i::h] = gh
            obj_terms.append((go, lw["lw_contact"] / h))
            stages["nn_idx"].append(idx)
            stages["con_obj"].append(go)
        terms.append((u, lw["lw_contact"] / h))
    rec = None
    if on("lw_inter"):
        rec = np.zeros((N, 8), f32)
        for i in range(h):
            rec[i::h] = inter_records(hands[i], vo, K)
        if model.optimize_object_scale:                 # the term reaches the (not detached) object then: sum over the hands
            gi = [((f32(0.0) - f32(lw["lw_inter"])) * np.ascontiguousarray(rec[i::h, 2:5]) / f32(Vo)) for i in range(h)]
            g = gi[0] + gi[1]
            obj_terms.append((np.ascontiguousarray(np.broadcast_to(g[:, None, :], (B, Vo, 3)), f32), 1.0))
    if dep is not None:
        obj_terms.append((dep[0], 1.0))
    return dict(vh=vh, vo=vo, mesh=mesh, terms=terms, rec=rec, obj_terms=obj_terms, stages=stages)


def _two_hand_param_grads(model, loss_weights, return_stages, two=None):
    lw = loss_weights
    on = lambda k: lw.get(k, 0.0) > 0
    h = 2
    c = lambda t: np.ascontiguousarray(t.detach().numpy(), f32)
    two = two_hand_terms(model, lw) if two is None else two
    N = two["vh"].shape[0]
    B = N // h
    pca = c(model.mano_pca_pose)
    P = pca.shape[1]
    arrs = [np.ascontiguousarray(t, f32) for t, _ in two["terms"]]
    ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    ws = np.asarray([w for _, w in two["terms"]] or [0.0], f32)
    g_frame = np.ascontiguousarray(two["rec"][:, 2:5]) if two["rec"] is not None else None
    g_mesh, g_r6, g_rt = np.empty((N, 778, 3), f32), np.zeros((N, 6), f32), np.zeros((N, 3), f32)
    # the hands' rigid backward as a launch of its own: one workgroup of 1024 threads per row (csrc/geometry.hip k_rigid_bwd<false>)
    clib.lib().orc_rigid_bwd_rows(clib.fptr(two["mesh"]), clib.fptr(c(model.rotations_hand).reshape(N, 6)),
                                  float(model.int_scales_hand.detach()[0]), ptrs, clib.fptr(ws), len(arrs),
                                  clib.fptr(g_frame) if g_frame is not None else None, 3,
                                  float(f32(lw["lw_inter"] / 778)) if g_frame is not None else 0.0, N, 778, 1024, clib.fptr(g_mesh),
                                  clib.fptr(g_r6), clib.fptr(g_rt))
    g_extra = np.ascontiguousarray((f32(2.0) * pca) * (f32(1.0) / f32(pca.size)), f32) if on("lw_pca") else None
    out = dict(mano_pca_pose=np.zeros((N, P), f32), mano_rot=np.zeros((N, 3), f32), mano_betas=np.zeros((N, 10), f32),
               mano_trans=np.zeros((N, 3), f32))
    rot, betas = c(model.mano_rot), c(model.mano_betas)
    for i, side in enumerate(model.hand_sides):
        lay = model.hands[side]["layout"]
        clib.lib().orc_mano_bwd_rows(*[clib.fptr(a) for a in lay[:7]], clib.iptr(lay[7]), clib.fptr(pca), P, clib.fptr(rot),
                                     clib.fptr(betas), clib.fptr(g_mesh), clib.fptr(g_extra) if g_extra is not None else None,
                                     float(lw.get("lw_pca", 0.0)), B, i, h, clib.fptr(out["mano_pca_pose"]),
                                     clib.fptr(out["mano_rot"]), clib.fptr(out["mano_betas"]), clib.fptr(out["mano_trans"]))
    out["rotations_hand"], out["translations_hand"] = g_r6.reshape(N, 3, 2), g_rt.reshape(N, 1, 3)
    if return_stages:
        return out, dict(two, g_mesh=g_mesh)
    return out


def hand_param_grads(model, loss_weights, return_stages=False, pair=None, depth_hand=None, two=None):
    """-> {name: float32 numpy array shaped like the parameter} for the six hand parameters (see the module docstring)."""
    lw = loss_weights
    on = lambda k: lw.get(k, 0.0) > 0
    if (model.hand_nb == 2 and (not on("lw_depth") or getattr(model, "ordinal_depth", False)) and not on("lw_sil_hand") and
            model.optimize_mano and
            isinstance(model.mano_betas, torch.nn.Parameter) and not model.int_scales_hand.requires_grad and
            model.losses.inter_type == "centroid"):
        return _two_hand_param_grads(model, lw, return_stages, two)
    inter_min = model.losses.inter_type == "min"
    if ((on("lw_depth") and depth_hand is None) or on("lw_sil_hand") or model.hand_nb != 1 or
            (model.optimize_mano and not isinstance(model.mano_betas, torch.nn.Parameter)) or
            model.int_scales_hand.requires_grad or model.losses.inter_type not in ("centroid", "min") or
            (inter_min and not model.optimize_mano)):
        raise NotImplementedError("the written-out hand chain covers the step-1 / step-2 loss sets of a one-hand clip")
    side = model.hand_sides[0]
    c = lambda t: np.ascontiguousarray(t.detach().numpy(), f32)
    with torch.no_grad():
        if model.optimize_mano:
            lbs_verts = model.mano_forward(model.mano_pca_pose, model.mano_rot, model.mano_betas, side)
            mesh_t = lbs_verts + model.mano_trans.unsqueeze(1)
        else:                                   # the hand mesh is an input (homan/homan.py:104-106): only its rigid pose moves
            mesh_t = model.verts_hand_og
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
    if on("lw_collision") or on("lw_contact"):
        pair = pair_terms(model, vh, vo, lw) if pair is None else pair
        if on("lw_collision"):
            terms.append((pair["col_hand"], lw["lw_collision"]))
        if on("lw_contact"):
            terms.append((pair["con_hand"], lw["lw_contact"]))
    if on("lw_depth"):          # d (lw_depth * loss_depth) / d hand vertices (oracle/depthchain.py), already times its weight
        terms.append((np.ascontiguousarray(depth_hand, f32), 1.0))
    rec = inter_records(vh, vo, K) if on("lw_inter") else None
    g_rigid = None
    if inter_min and on("lw_inter"):
        # inter_type "min" (homan/losses.py:219-221): on the frames the gate lets through, the smallest squared distance between a
        # hand and an object vertex; the pair is named by the search (first minimum), the pull acts on the hand's vertex - rigid
        # pose only, the mesh is detached there (homan/homan.py:482-490)
        g_rigid, min_stage = min_pair_pull(vh, vo, rec, lw["lw_inter"])
    if not model.optimize_mano:
        # rigid pose only: the launch of its own (csrc/geometry.hip k_rigid_bwd<false>, one 1024-thread workgroup per frame)
        arrs = [np.ascontiguousarray(t, f32) for t, _ in terms]
        ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        ws = np.asarray([w for _, w in terms] or [0.0], f32)
        g_frame = np.ascontiguousarray(rec[:, 2:5]) if rec is not None else None
        g_mesh, g_r6, g_rt = np.empty((B, 778, 3), f32), np.zeros((B, 6), f32), np.zeros((B, 3), f32)
        clib.lib().orc_rigid_bwd_rows(clib.fptr(mesh), clib.fptr(c(model.rotations_hand).reshape(B, 6)),
                                      float(model.int_scales_hand.detach()[0]), ptrs, clib.fptr(ws), len(arrs),
                                      clib.fptr(g_frame) if g_frame is not None else None, 3,
                                      float(f32(lw["lw_inter"] / 778)) if g_frame is not None else 0.0, B, 778, 1024,
                                      clib.fptr(g_mesh), clib.fptr(g_r6), clib.fptr(g_rt))
        out = dict(rotations_hand=g_r6.reshape(B, 3, 2), translations_hand=g_rt.reshape(B, 1, 3))
        if return_stages:
            return out, dict(mesh=mesh, vh=vh, vo=vo, terms=terms, rec=rec, pair=pair)
        return out
    pca = c(model.mano_pca_pose)
    P = pca.shape[1]
    g_extra = None
    if on("lw_pca"):
        g_extra = np.ascontiguousarray((f32(2.0) * pca) * (f32(1.0) / f32(pca.size)), f32)
    arrs = [np.ascontiguousarray(t, f32) for t, _ in terms]
    ptrs = (ctypes.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    ws = np.asarray([w for _, w in terms] or [0.0], f32)
    g_frame = np.ascontiguousarray(rec[:, 2:5]) if (rec is not None and not inter_min) else None
    out = dict(mano_pca_pose=np.zeros((B, P), f32), mano_rot=np.zeros((B, 3), f32), mano_betas=np.zeros((B, 10), f32),
               mano_trans=np.zeros((B, 3), f32), rotations_hand=np.zeros((B, 6), f32), translations_hand=np.zeros((B, 3), f32))
    lay = model.hands[side]["layout"]
    if g_rigid is not None:
        clib.lib().orc_hand_chain_rigid(*[clib.fptr(a) for a in lay[:7]], clib.iptr(lay[7]), clib.fptr(pca), P,
                                        clib.fptr(c(model.mano_rot)), clib.fptr(c(model.mano_betas)), clib.fptr(mesh),
                                        clib.fptr(c(model.rotations_hand).reshape(B, 6)), float(model.int_scales_hand.detach()[0]),
                                        ptrs, clib.fptr(ws), len(arrs), clib.fptr(g_rigid),
                                        clib.fptr(g_extra) if g_extra is not None else None, float(lw.get("lw_pca", 0.0)), B,
                                        clib.fptr(out["mano_pca_pose"]), clib.fptr(out["mano_rot"]), clib.fptr(out["mano_betas"]),
                                        clib.fptr(out["mano_trans"]), clib.fptr(out["rotations_hand"]),
                                        clib.fptr(out["translations_hand"]))
        out["rotations_hand"] = out["rotations_hand"].reshape(B, 3, 2)
        out["translations_hand"] = out["translations_hand"].reshape(B, 1, 3)
        if return_stages:
            return out, dict(mesh=mesh, vh=vh, vo=vo, terms=terms, rec=rec, pair=pair, g_rigid=g_rigid, min_pair=min_stage)
        return out
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
        return out, dict(mesh=mesh, vh=vh, vo=vo, terms=terms, rec=rec, pair=pair)
    return out
