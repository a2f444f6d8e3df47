"""The object-pose initialisation's step in ONE written-out evaluation order (TEST INFRASTRUCTURE -- see oracle/__init__.py).

`pose_grads(model)` evaluates, for an `oracle.poseopt.PoseOptimizer(written_out=True)` (lw_chamfer = 0: the reference's only call
site) at its current candidate poses, the per-candidate losses and the gradients of their sum with respect to `rotations` and
`translations` (reference homan/pose_optimization.py:98-160: rigid transform, occlusion-aware silhouette L2 on the rasteriser's
un-pooled sample grid, off-screen penalty) as one fixed sequence of IEEE fp32 operations: the silhouette's pseudo-gradient and the
rigid backward with order-independent sums on the grid 2^-24 (oracle/csrc/objchain.c), the off-screen term's value in the order of
csrc/losses.hip (oracle/csrc/lbs_exact.c orc_offscreen).  `reproducible_fit(model, lr, steps)` is the loop of
pose_optimization.py:330-357 on top of it with the written-out Adam (oracle/adam.py) and the best-ever bookkeeping: bit-equal to
homan_amd.pose_optimization's fused loop.  Same mathematics as autograd through PoseOptimizer.forward (tests/test_poseinit.py).
"""
import numpy as np
import torch

from . import clib
from . import nmr as o_nmr
from .adam import Adam
from .objchain import build_adjacency, pseudo_gradient_exact, rigid_bwd_sil_exact

f32 = np.float32
SUM_LOG2Q = -24           # grid of the exact sums (homan_amd/pose_optimization.py: unnormalised sums of squares)
OFFSCREEN_WEIGHT = 100000.0
BLOCK_THREADS = 256


def pose_grads(model):
    """-> ({"rotations": (n,3,2), "translations": (n,1,3)} float32, losses (n,) float32 = mask + offscreen)"""
    assert model.written_out and model.lw_chamfer == 0
    size = int(model.image_size)
    with torch.no_grad():
        verts_t = model.apply_transformation()
        rend = o_nmr.Renderer(image_size=size, K=model.K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1,
                              anti_aliasing=False)
        faces_t = rend._ndc_faces(verts_t, model.faces, None, None, None, None, None)
    verts = np.ascontiguousarray(verts_t.numpy(), f32)
    n, V = verts.shape[:2]
    F = model.faces.shape[1]
    ndc = np.ascontiguousarray(faces_t.numpy().reshape(n, 2 * F, 9), f32)
    idx = np.empty((n, size, size), np.int32)
    dep = np.empty((n, size, size), f32)
    clib.lib().orc_nmr_face_index_map(clib.fptr(ndc), n, 2 * F, size, rend.near, rend.far, clib.iptr(idx), clib.fptr(dep))
    keep, ref = model.keep_mask[0].numpy().astype(f32), model.image_ref[0].numpy().astype(f32)
    alpha = (idx >= 0).astype(f32)[:, ::-1]                                  # the rendered image (vertical flip)
    dimg = keep[None] * (keep[None] * alpha - ref[None])
    mask_loss = (dimg * dimg).reshape(n, -1).sum(1, dtype=np.float64).astype(f32)        # a count of samples: exact
    ga = np.ascontiguousarray((f32(2.0) * dimg)[:, ::-1], f32)                # d sum((image - ref)^2) / d sample, un-flipped grid
    parts = pseudo_gradient_exact(ndc, idx, ga, F, rend.rasterizer_eps, SUM_LOG2Q)
    K1 = np.ascontiguousarray(model.K[0].numpy(), f32)
    off, g_off = np.empty(n, f32), np.empty((n, V, 3), f32)
    clib.lib().orc_offscreen(clib.fptr(verts), clib.fptr(K1), n, V, float(o_nmr.DEFAULT_FAR), OFFSCREEN_WEIGHT, BLOCK_THREADS,
                             clib.fptr(off), clib.fptr(g_off))
    adj = build_adjacency(model.faces[0].numpy(), V)
    mesh = np.ascontiguousarray(model.vertices.numpy(), f32)
    rot6d = np.ascontiguousarray(model.rotations.detach().numpy().reshape(n, 6), f32)
    K_all = np.ascontiguousarray(np.broadcast_to(K1[None], (n, 3, 3)), f32)
    g_rot, g_tr, _, _ = rigid_bwd_sil_exact(mesh, rot6d, 1.0, 0, [(g_off, 1.0)], parts, adj, verts, K_all, 1.0, F, SUM_LOG2Q)
    return {"rotations": g_rot.reshape(n, 3, 2), "translations": g_tr.reshape(n, 1, 3)}, mask_loss + off


def reproducible_fit(model, lr, num_iterations):
    """-> (final losses (n,), best-ever rotation (3,2), best-ever translation (1,3)); the candidates stay in `model`"""
    params = [model.rotations, model.translations]
    opt = Adam([{"params": params, "lr": lr}])
    best, best_rot, best_trans = np.float32(np.inf), None, None
    losses = None
    for _ in range(num_iterations):
        grads, losses = pose_grads(model)
        for p, k in zip(params, ("rotations", "translations")):
            p.grad = torch.from_numpy(grads[k]).reshape(p.shape)
        opt.step()
        if not np.isnan(losses).any() and losses.min() < best:        # the pose is copied AFTER the step (:348-353), strict <
            ind = int(np.argmin(losses))                               # (first minimum)
            best = losses[ind]
            best_rot, best_trans = model.rotations[ind].detach().clone(), model.translations[ind].detach().clone()
    return losses, best_rot, best_trans
