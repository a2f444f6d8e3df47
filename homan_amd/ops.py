"""Autograd-facing wrappers over the HIP kernels (C ABI via ctypes).

Each op mirrors one leaf the reference calls (file:line in the docstrings).  Forward and
backward arithmetic is entirely in libhoman_amd.so; torch supplies memory, streams and the
autograd graph.
"""
import numpy as np
import torch

from . import lib as _lib

NMR_NEAR, NMR_FAR, NMR_EPS = 0.1, 100.0, 1e-3   # neural_renderer ctor defaults used at reference losses.py:73-77


def _f32(t):
    return t.contiguous().float()


def build_adjacency(faces_np, num_verts):
    """CSR vertex -> (face*3 + corner) lists for a (F,3) face array (host, once per topology)."""
    faces_np = np.asarray(faces_np).astype(np.int64)
    flat = faces_np.reshape(-1)
    order = np.argsort(flat, kind="stable")
    counts = np.bincount(flat, minlength=num_verts)
    off = np.zeros(num_verts + 1, np.int32)
    off[1:] = np.cumsum(counts)
    return torch.from_numpy(off), torch.from_numpy(order.astype(np.int32))


class SilhouetteContext:
    """Per-call-site state of the silhouette op: topology adjacency + the workspace that carries
    the forward intermediates (packed faces, index map, bit planes) to the backward."""

    def __init__(self, faces, num_verts, batch, size, device):
        assert faces.dim() == 3 and faces.shape[0] == batch
        f0 = faces[0].detach().cpu().numpy()
        if batch > 1:
            assert bool((faces == faces[:1]).all()), "per-frame topologies must be identical"
        self.B, self.V, self.F, self.S = batch, num_verts, f0.shape[0], size
        self.faces = faces[0].to(device=device, dtype=torch.int32).contiguous()
        off, items = build_adjacency(f0, num_verts)
        self.adj_off, self.adj_items = off.to(device), items.to(device)
        nbytes = _lib.lib().hm_sil_workspace_bytes(self.B, self.V, self.F, self.S)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def idx_map(self):
        out = torch.empty(self.B, 2 * self.S, 2 * self.S, dtype=torch.int32, device=self.workspace.device)
        _lib.check(_lib.lib().hm_sil_read_idx_map(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S,
                                                  _lib.ptr(out), _lib.stream()), "hm_sil_read_idx_map")
        return out

    def faces9(self):
        out = torch.empty(self.B, self.F, 9, device=self.workspace.device)
        _lib.check(_lib.lib().hm_sil_read_faces9(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S,
                                                 _lib.ptr(out), _lib.stream()), "hm_sil_read_faces9")
        return out


class _SilhouetteLoss(torch.autograd.Function):
    """reference homan/losses.py:183-197 (render @ ROI intrinsics, keep-masked MSE, /sum(keep), /B, IoU metric)."""

    @staticmethod
    def forward(ctx, verts, K, keep, ref, keep_sum, sctx, orig_size):
        verts, K = _f32(verts), _f32(K)
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        out = torch.empty(2, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, _lib.ptr(keep), _lib.ptr(ref), _lib.ptr(keep_sum),
            _lib.ptr(pooled), _lib.ptr(out), _lib.ptr(sctx.workspace), _lib.stream()), "hm_sil_fwd")
        ctx.save_for_backward(verts, K, keep_sum)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        ctx.mark_non_differentiable(pooled)
        return out[0:1], out[1], pooled

    @staticmethod
    def backward(ctx, g_loss, _g_iou, _g_img):
        verts, K, keep_sum = ctx.saved_tensors
        sctx = ctx.sctx
        g_loss = _f32(g_loss).reshape(1)
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), NMR_EPS, 1,
            _lib.ptr(g_loss), None, _lib.ptr(keep_sum), _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(grad_verts), None, _lib.ptr(sctx.workspace), _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None, None, None, None


class _SilhouetteRender(torch.autograd.Function):
    """reference homan/losses.py:187: renderer(verts, faces, K=, mode='silhouettes') -> (B,S,S)."""

    @staticmethod
    def forward(ctx, verts, K, sctx, orig_size):
        verts, K = _f32(verts), _f32(K)
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, None, None, None, _lib.ptr(pooled), None,
            _lib.ptr(sctx.workspace), _lib.stream()), "hm_sil_fwd")
        ctx.save_for_backward(verts, K)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        return pooled

    @staticmethod
    def backward(ctx, g_img, grad_ndc_out=None):
        verts, K = ctx.saved_tensors
        sctx = ctx.sctx
        g_img = _f32(g_img)
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), NMR_EPS, 0,
            None, _lib.ptr(g_img), None, _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(grad_verts), _lib.ptr(sctx.grad_ndc) if getattr(sctx, "grad_ndc", None) is not None else None,
            _lib.ptr(sctx.workspace), _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None


def silhouette_loss(verts, K, keep, ref, keep_sum, sctx, orig_size=1.0):
    """-> (loss_sil (1,), mean IoU (0-d), silhouettes (B,S,S))."""
    return _SilhouetteLoss.apply(verts, K, keep, ref, keep_sum, sctx, orig_size)


def silhouette_render(verts, K, sctx, orig_size=1.0):
    return _SilhouetteRender.apply(verts, K, sctx, orig_size)
