"""The ordinal depth term's gradient on the two meshes' vertices in ONE written-out evaluation order (TEST INFRASTRUCTURE -- see
oracle/__init__.py).

`depth_vertex_grads(model, lw_depth)` -> d (lw_depth * loss_depth) / d camera-space vertices of the object (B,Vo,3) and of the hand
(B,778,3) for an `oracle.model.OracleHOMan` built with ordinal_depth=True (one hand).  Same mathematics as autograd through
`OracleHOMan.compute_ordinal_depth_loss` (reference homan/homan.py:384-419 + lossutils.py:133-169 as the method intends - the
reference's own call site raises); the order of every sum is the one csrc/raster.hip uses (oracle/csrc/lbs_exact.c:
orc_ordinal_depth_grad, orc_depth_bwd_faces, orc_depth_bwd_gather), the logistic function is the shared hm_sigmoid.

Chain:  depth + coverage renders of both meshes at the full-image camera (oracle.nmr: hard rasteriser, 2x2 samples per pixel,
z-buffer, vertical flip, average pool in the order ((s00 + s01) + s10) + s11)  ->  per-pixel gradient of the ordinal term  ->
NMR's depth-map backward per (face, winding)  ->  vertex gather + projection backward.
"""
import numpy as np
import torch

from . import clib
from . import nmr as o_nmr
from .objchain import build_adjacency

f32 = np.float32


def render_layers(model, verts, faces):
    """(B,V,3) camera-space vertices, faces (F,3) -> NDC faces (B,F,9), owner map (B,is,is), pooled depth (B,S,S), full-coverage
    mask (B,S,S) uint8 at the full-image camera (Renderer(image_size, K=camintr, orig_size=1), homan/homan.py:168-172)"""
    S = int(model.image_size)
    rend = o_nmr.Renderer(image_size=S, K=model.camintr, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    B, F = verts.shape[0], faces.shape[0]
    with torch.no_grad():
        ndc = rend._ndc_faces(verts, faces[None].expand(B, -1, -1), None, None, None, None, None)
    ndc = np.ascontiguousarray(ndc.numpy().reshape(B, 2 * F, 9), f32)
    idx = np.empty((B, 2 * S, 2 * S), np.int32)
    dep = np.empty((B, 2 * S, 2 * S), f32)
    clib.lib().orc_nmr_face_index_map(clib.fptr(ndc), B, 2 * F, 2 * S, rend.near, rend.far, clib.iptr(idx), clib.fptr(dep))
    d = dep[:, ::-1]                                   # vertical flip, then the four samples of a pixel in a stated order
    pooled = (((d[:, 0::2, 0::2] + d[:, 0::2, 1::2]) + d[:, 1::2, 0::2]) + d[:, 1::2, 1::2]) / f32(4.0)
    c = (idx >= 0)[:, ::-1]
    full = c[:, 0::2, 0::2] & c[:, 0::2, 1::2] & c[:, 1::2, 0::2] & c[:, 1::2, 1::2]
    return (np.ascontiguousarray(ndc[:, :F]), idx, np.ascontiguousarray(pooled, f32), np.ascontiguousarray(full.astype(np.uint8)))


def depth_vertex_grads(model, lw_depth, return_stages=False):
    if model.hand_nb != 1 or not getattr(model, "ordinal_depth", False):
        raise NotImplementedError("the written-out depth chain: one hand, a model built with ordinal_depth=True")
    S = int(model.image_size)
    with torch.no_grad():
        vo, vh = model.get_verts_object()[0], model.get_verts_hand()[0]
    B = vo.shape[0]
    fo, fh = model.faces_object[0], model.faces_hand[0]
    lay = [render_layers(model, vo, fo), render_layers(model, vh, fh)]
    m0 = np.ascontiguousarray((model.masks_object != 0).numpy().astype(np.uint8))
    m1 = np.ascontiguousarray((model.masks_human != 0).numpy().astype(np.uint8))
    assert m0.shape == (B, S, S) and m1.shape == (B, S, S)
    g = [np.empty((B, S, S), f32), np.empty((B, S, S), f32)]
    rec = np.zeros(8, f32)
    clib.lib().orc_ordinal_depth_grad(clib.fptr(lay[0][2]), clib.fptr(lay[1][2]), clib.u8ptr(lay[0][3]), clib.u8ptr(lay[1][3]),
                                      clib.u8ptr(m0), clib.u8ptr(m1), B, S, float(f32(lw_depth)), clib.fptr(g[0]), clib.fptr(g[1]),
                                      clib.fptr(rec))
    K = np.ascontiguousarray(model.camintr.numpy(), f32)
    out = []
    for (ndc, idx, _, _), gp, verts, faces in zip(lay, g, (vo, vh), (fo, fh)):
        F, V = faces.shape[0], verts.shape[1]
        gf9 = np.empty((B, F, 2, 9), f32)
        clib.lib().orc_depth_bwd_faces(clib.fptr(ndc), clib.iptr(idx), clib.fptr(gp), B, F, S, clib.fptr(gf9))
        adj = build_adjacency(faces.numpy(), V)
        gv = np.empty((B, V, 3), f32)
        v = np.ascontiguousarray(verts.numpy(), f32)
        clib.lib().orc_depth_bwd_gather(clib.fptr(gf9), clib.iptr(adj[0]), clib.iptr(adj[1]), clib.fptr(v), clib.fptr(K), B, V, F,
                                        1.0, clib.fptr(gv))
        out.append(gv)
    if return_stages:
        return out[0], out[1], dict(pooled=(lay[0][2], lay[1][2]), full=(lay[0][3], lay[1][3]), g=g, rec=rec)
    return out[0], out[1]
