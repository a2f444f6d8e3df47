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
        # The kernels tile the image in 32x32-pixel blocks (64-sample mask words of the edge sweeps).  Any other image size
        # (the reference's Core50 setting renders at 350, homan/getdataset.py:35) is rendered on the next multiple of 32
        # with the first two rows of K scaled by size / padded size: the top-left size x size block of that render shows
        # exactly the rays of the requested image (same pinhole, coarser normalisation), the rest is cropped away.  Sample
        # positions then round differently from a native render of that size: coverage agrees up to projection rounding
        # (tests/test_render_gpu.py), where multiples of 32 are bit-exact against the oracle.
        self.size = int(size)
        size = (self.size + 31) // 32 * 32
        self.padded = size != self.size
        self.B, self.V, self.F, self.S = batch, num_verts, f0.shape[0], size
        self.faces = faces[0].to(device=device, dtype=torch.int32).contiguous()
        off, items = build_adjacency(f0, num_verts)
        self.adj_off, self.adj_items = off.to(device), items.to(device)
        # dispatch order of the forward raster's (frame, 32x32-sample region) workgroups: default = regions from the
        # ROI centre outwards, frame fastest; `calibrate()` replaces it by a cost-sorted order
        n = size // 16
        ry, rx = np.divmod(np.arange(n * n), n)
        ring = np.maximum(np.abs(2 * rx + 1 - n), np.abs(2 * ry + 1 - n))
        regions = np.argsort(ring, kind="stable").astype(np.int64)
        wo = (np.arange(batch, dtype=np.int64)[None, :] << 16) | regions[:, None]
        self.work_order = torch.from_numpy(wo.reshape(-1).astype(np.int32)).to(device)
        self.face_order = None
        # grid of the backward's order-independent sums (include/homan_amd.h, "ORDER-INDEPENDENT SUMS"): 0 = the default
        # 2^-44, right for the normalised silhouette loss of the joint fit; the pose initialisation's unnormalised sums of
        # squares set a coarser one
        self.sum_log2q = 0
        nbytes = _lib.lib().hm_sil_workspace_bytes(self.B, self.V, self.F, self.S)
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=device)   # holds a self-resetting ticket

    # ---- image sizes that are not a multiple of 32 (see __init__)
    def K_eff(self, K):
        if not self.padded:
            return K
        K = K.clone()
        K[:, :2, :] *= self.size / self.S
        return K.contiguous()

    def crop(self, img):
        return img[..., :self.size, :self.size].contiguous() if self.padded else img

    def pad(self, img, value=0.0):
        if not self.padded:
            return img
        d = self.S - self.size
        return torch.nn.functional.pad(img, (0, d, 0, d), value=value).contiguous()

    def eps(self):
        """the rasteriser's eps (NDC units, added to every pseudo-distance of the backward) in the units of the grid the
        kernels render on: a padded render's NDC is the requested image's times size / S, and d loss / d NDC is carried
        back through that factor, so eps scales with it (otherwise the gradient is off by ~eps / pixel pitch: 1-2 %)"""
        return NMR_EPS * self.size / self.S if self.padded else NMR_EPS

    def crop_samples(self, img):
        """(.., 2S, 2S) sample-resolution image -> the (.., 2 size, 2 size) block the caller asked for (no-AA renders)"""
        n = 2 * self.size
        return img[..., :n, :n].contiguous() if self.padded else img

    def pad_samples(self, img, value=0.0):
        if not self.padded:
            return img
        d = 2 * (self.S - self.size)
        return torch.nn.functional.pad(img, (0, d, 0, d), value=value).contiguous()

    def calibrate(self):
        """Cost-sorted launch orders from the screen boxes of the last forward (poses move little during an
        optimisation, so the statistics of the current state predict the cost of the next iterations):
        forward raster workgroups by descending candidate count.
        Pure scheduling: results do not depend on the order."""
        raw = torch.empty(self.B * self.F * 8, dtype=torch.uint8, device=self.workspace.device)
        _lib.check(_lib.lib().hm_sil_read_boxes(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S, _lib.ptr(raw),
                                                _lib.stream()), "hm_sil_read_boxes")
        bx = raw.cpu().numpy().view(np.uint16).reshape(self.B, self.F, 4).astype(np.int64)
        valid = (bx[..., 0] >> 14) != 0
        x0, y0, x1, y1 = bx[..., 0] & 0x3fff, bx[..., 1], bx[..., 2], bx[..., 3]
        n, is_ = self.S // 16, 2 * self.S
        # region index of a sample: columns x // 32 ; rows are flipped (tile row 0 holds the largest yi)
        rx0, rx1 = x0 // 32, x1 // 32
        ry0, ry1 = (is_ - 1 - y1) // 32, (is_ - 1 - y0) // 32
        cnt = np.zeros((self.B, n + 1, n + 1), np.int64)
        bi = np.broadcast_to(np.arange(self.B)[:, None], x0.shape)[valid]
        a0, a1, c0, c1 = ry0[valid], ry1[valid] + 1, rx0[valid], rx1[valid] + 1
        np.add.at(cnt, (bi, a0, c0), 1)
        np.add.at(cnt, (bi, a1, c1), 1)
        np.add.at(cnt, (bi, a0, c1), -1)
        np.add.at(cnt, (bi, a1, c0), -1)
        cnt = cnt.cumsum(1).cumsum(2)[:, :n, :n].reshape(self.B, n * n)
        order = np.argsort(-cnt.reshape(-1), kind="stable")
        b_idx, region = np.divmod(order, n * n)
        self.work_order = torch.from_numpy(((b_idx << 16) | region).astype(np.int32)).to(self.workspace.device)
        # winding class that owns the samples (the camera-facing surface of a closed mesh): rasterised first, the other
        # class is then rejected block-wise behind it
        idx = self.idx_map()
        near = int(((idx >= self.F).sum() > ((idx >= 0) & (idx < self.F)).sum()).item())
        _lib.check(_lib.lib().hm_sil_hint_near_winding(_lib.ptr(self.workspace), near, _lib.stream()),
                   "hm_sil_hint_near_winding")
        self.near_winding = near
        # (the edge sweeps keep the natural face order: sorting faces by box perimeter was measured slower -- it scatters
        #  neighbouring faces, and with them the cache lines of the index map and the mask planes they share)
        self.face_order = None

    def idx_map(self):
        out = torch.empty(self.B, 2 * self.S, 2 * self.S, dtype=torch.int32, device=self.workspace.device)
        _lib.check(_lib.lib().hm_sil_read_idx_map(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S,
                                                  _lib.ptr(out), _lib.stream()), "hm_sil_read_idx_map")
        return out

    def invalidate_outputs(self):
        """the loss inputs (keep / ref masks) in the caller's buffers changed: see hm_sil_invalidate_outputs"""
        _lib.check(_lib.lib().hm_sil_invalidate_outputs(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S,
                                                        _lib.stream()), "hm_sil_invalidate_outputs")

    def parts(self):
        """(B,F,3,2) float64: per-(face, corner) NDC gradients of the last backward (exact sums, see include/homan_amd.h)"""
        out = torch.empty(self.B, self.F, 3, 2, dtype=torch.float64, device=self.workspace.device)
        _lib.check(_lib.lib().hm_sil_read_parts(_lib.ptr(self.workspace), self.B, self.V, self.F, self.S,
                                                _lib.ptr(out), _lib.stream()), "hm_sil_read_parts")
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
        verts, K = _f32(verts), sctx.K_eff(_f32(K))
        keep, ref = sctx.pad(keep), sctx.pad(ref)          # (padding: keep = 0, so it counts for nothing)
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        out = torch.empty(2, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, _lib.ptr(keep), _lib.ptr(ref), _lib.ptr(keep_sum),
            _lib.ptr(pooled), _lib.ptr(out), _lib.ptr(sctx.work_order), None, None, 0, None, None, None, 0, 0,
            _lib.ptr(sctx.workspace), _lib.stream()),
            "hm_sil_fwd")
        ctx.save_for_backward(verts, K, keep_sum)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        pooled = sctx.crop(pooled)
        ctx.mark_non_differentiable(pooled)
        return out[0:1], out[1], pooled

    @staticmethod
    def backward(ctx, g_loss, _g_iou, _g_img):
        verts, K, keep_sum = ctx.saved_tensors
        sctx = ctx.sctx
        g_loss = _f32(g_loss).reshape(1)
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), sctx.eps(), 1,
            _lib.ptr(g_loss), None, _lib.ptr(keep_sum), _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(sctx.face_order), _lib.ptr(grad_verts), None, _lib.ptr(sctx.workspace), sctx.sum_log2q, _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None, None, None, None


class _SilhouetteRender(torch.autograd.Function):
    """reference homan/losses.py:187: renderer(verts, faces, K=, mode='silhouettes') -> (B,S,S)."""

    @staticmethod
    def forward(ctx, verts, K, sctx, orig_size):
        verts, K = _f32(verts), sctx.K_eff(_f32(K))
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, None, None, None, _lib.ptr(pooled), None,
            _lib.ptr(sctx.work_order), None, None, 0, None, None, None, 0, 0, _lib.ptr(sctx.workspace), _lib.stream()),
            "hm_sil_fwd")
        ctx.save_for_backward(verts, K)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        return sctx.crop(pooled)

    @staticmethod
    def backward(ctx, g_img, grad_ndc_out=None):
        verts, K = ctx.saved_tensors
        sctx = ctx.sctx
        g_img = sctx.pad(_f32(g_img))
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), sctx.eps(), 0,
            None, _lib.ptr(g_img), None, _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(sctx.face_order), _lib.ptr(grad_verts), _lib.ptr(sctx.grad_ndc) if getattr(sctx, "grad_ndc", None) is not None else None,
            _lib.ptr(sctx.workspace), sctx.sum_log2q, _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None


class _SilhouetteRenderNoAA(torch.autograd.Function):
    """reference homan/pose_optimization.py:89-96,140-141: nr.Renderer(image_size, anti_aliasing=False)(verts, faces,
    mode='silhouettes') -> (B,image_size,image_size) hard 0/1 coverage.  `sctx` is built with size = image_size // 2
    (its sample grid IS the image); the backward is the same edge-sweep pseudo-gradient, fed per sample."""

    @staticmethod
    def forward(ctx, verts, K, sctx, orig_size):
        verts, K = _f32(verts), sctx.K_eff(_f32(K))       # (sizes off the 64-sample grid: padded render, cropped below)
        n = 2 * sctx.S
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        alpha = torch.empty(sctx.B, n, n, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, None, None, None, _lib.ptr(pooled), None,
            _lib.ptr(sctx.work_order), None, _lib.ptr(alpha), 0, None, None, None, 0, 0, _lib.ptr(sctx.workspace),
            _lib.stream()), "hm_sil_fwd")
        ctx.save_for_backward(verts, K)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        return sctx.crop_samples(alpha)

    @staticmethod
    def backward(ctx, g_img):
        verts, K = ctx.saved_tensors
        sctx = ctx.sctx
        g_img = sctx.pad_samples(_f32(g_img))
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), sctx.eps(), 3,
            None, _lib.ptr(g_img), None, _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(sctx.face_order), _lib.ptr(grad_verts), None, _lib.ptr(sctx.workspace), sctx.sum_log2q, _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None


def silhouette_render_noaa(verts, K, sctx, orig_size=1.0):
    """Hard silhouettes without anti-aliasing, (B, 2*sctx.S, 2*sctx.S)."""
    return _SilhouetteRenderNoAA.apply(verts, K, sctx, orig_size)


class _MaskedSilhouetteL2NoAA(torch.autograd.Function):
    """reference homan/pose_optimization.py:138-143 in one pass: image = keep * Renderer(anti_aliasing=False)(verts) ;
    loss_b = sum((image_b - ref)^2) ; iou_b.  keep / ref: one (n,n) mask shared by all poses.  The backward reuses the
    sweep planes the forward emitted, which is valid for POSITIVE upstream gradients only (a sum of per-pose losses)."""

    @staticmethod
    def forward(ctx, verts, K, keep, ref, sctx, orig_size):
        verts, K = _f32(verts), sctx.K_eff(_f32(K))
        n = 2 * sctx.S
        keep, ref = sctx.pad_samples(keep), sctx.pad_samples(ref)      # (padding: keep = 0, it counts for nothing)
        assert keep.shape == (n, n) and ref.shape == (n, n) and keep.is_contiguous() and ref.is_contiguous()
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        alpha = torch.empty(sctx.B, n, n, device=verts.device)
        frame = torch.empty(sctx.B, 2, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, _lib.ptr(keep), _lib.ptr(ref), None, _lib.ptr(pooled), None,
            _lib.ptr(sctx.work_order), None, _lib.ptr(alpha), 1, None, None, None, 0, 0, _lib.ptr(sctx.workspace),
            _lib.stream()), "hm_sil_fwd")
        _lib.check(_lib.lib().hm_sil_reduce(sctx.B, sctx.V, sctx.F, sctx.S, None, None, _lib.ptr(frame),
                                            _lib.ptr(sctx.workspace), _lib.stream()), "hm_sil_reduce")
        ctx.save_for_backward(verts, K)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        alpha = sctx.crop_samples(alpha)
        ctx.mark_non_differentiable(alpha)
        return frame[:, 0], frame[:, 1], alpha

    @staticmethod
    def backward(ctx, g_loss, _g_iou, _g_alpha):
        verts, K = ctx.saved_tensors
        sctx = ctx.sctx
        g_loss = _f32(g_loss).contiguous()
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_sil_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size), sctx.eps(), 4,
            _lib.ptr(g_loss), None, None, _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items),
            _lib.ptr(sctx.face_order), _lib.ptr(grad_verts), None, _lib.ptr(sctx.workspace), sctx.sum_log2q, _lib.stream()), "hm_sil_bwd")
        return grad_verts, None, None, None, None, None


def masked_silhouette_l2_noaa(verts, K, keep, ref, sctx, orig_size=1.0):
    """-> (per-pose sum of squares (B,), per-pose IoU (B,), coverage image (B,n,n) [no gradient])."""
    return _MaskedSilhouetteL2NoAA.apply(verts, K, keep, ref, sctx, orig_size)


def silhouette_loss(verts, K, keep, ref, keep_sum, sctx, orig_size=1.0):
    """-> (loss_sil (1,), mean IoU (0-d), silhouettes (B,S,S))."""
    return _SilhouetteLoss.apply(verts, K, keep, ref, keep_sum, sctx, orig_size)


def silhouette_render(verts, K, sctx, orig_size=1.0):
    return _SilhouetteRender.apply(verts, K, sctx, orig_size)


class _DepthRender(torch.autograd.Function):
    """reference homan/homan.py:391,406: `_, depths, sils = renderer.render(verts, faces, textures, K=)`.
    -> (silhouettes (B,S,S), depths (B,S,S)); only the depth image is differentiable here (the ordinal depth loss uses
    the silhouettes as a mask only: lossutils.py:146-149 compares them with ==)."""

    @staticmethod
    def forward(ctx, verts, K, sctx, orig_size):
        verts, K = _f32(verts), sctx.K_eff(_f32(K))
        pooled = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        depth = torch.empty(sctx.B, sctx.S, sctx.S, device=verts.device)
        _lib.check(_lib.lib().hm_sil_fwd(
            _lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S,
            float(orig_size), NMR_NEAR, NMR_FAR, None, None, None, _lib.ptr(pooled), None,
            _lib.ptr(sctx.work_order), _lib.ptr(depth), None, 0, None, None, None, 0, 0, _lib.ptr(sctx.workspace),
            _lib.stream()),
            "hm_sil_fwd")
        ctx.save_for_backward(verts, K)
        ctx.sctx, ctx.orig_size = sctx, orig_size
        pooled, depth = sctx.crop(pooled), sctx.crop(depth)
        ctx.mark_non_differentiable(pooled)
        return pooled, depth

    @staticmethod
    def backward(ctx, _g_sil, g_depth):
        verts, K = ctx.saved_tensors
        sctx = ctx.sctx
        grad_verts = torch.empty_like(verts)
        _lib.check(_lib.lib().hm_depth_bwd(
            _lib.ptr(verts), _lib.ptr(K), sctx.B, sctx.V, sctx.F, sctx.S, float(ctx.orig_size),
            _lib.ptr(sctx.pad(_f32(g_depth))),
            _lib.ptr(sctx.adj_off), _lib.ptr(sctx.adj_items), _lib.ptr(grad_verts), _lib.ptr(sctx.workspace),
            _lib.stream()), "hm_depth_bwd")
        return grad_verts, None, None, None


def depth_render(verts, K, sctx, orig_size):
    """-> (silhouettes, depths), both (B,S,S).  One SilhouetteContext per rendered mesh (the backward reads the
    forward's index map from its workspace)."""
    return _DepthRender.apply(verts, K, sctx, orig_size)


def render_rgbd(verts, K, sctx, textures, light_direction=(0, 1, 0), intensity_ambient=0.5, intensity_directional=0.5,
                background_color=(0, 0, 0), orig_size=1.0):
    """`renderer.render(vertices, faces, textures, K=)` -> (rgb (B,3,S,S), depth (B,S,S), alpha (B,S,S)) for the
    reference's per-face colour textures (B,F,1,1,1,3) (reference homan/homan.py:510-545).  Visualisation only: no
    gradient.  Rasterises with csrc/raster_*.hip (hm_sil_fwd) and shades its index map (hm_shade_rgb)."""
    import ctypes
    with torch.no_grad():
        verts, K = _f32(verts.detach()), _f32(K)
        alpha, depth = _DepthRender.apply(verts, K, sctx, orig_size)      # (cropped to the requested size)
        tex = _f32(textures.reshape(sctx.B, sctx.F, 3))
        rgb = torch.empty(sctx.B, 3, sctx.S, sctx.S, device=verts.device)
        ld = (ctypes.c_float * 3)(*[float(x) for x in light_direction])
        bg = (ctypes.c_float * 3)(*[float(x) for x in background_color])
        _lib.check(_lib.lib().hm_shade_rgb(_lib.ptr(verts), _lib.ptr(sctx.faces), 0, _lib.ptr(tex), sctx.B, sctx.V,
                                           sctx.F, sctx.S, ctypes.cast(ld, ctypes.c_void_p),
                                           float(intensity_ambient), float(intensity_directional),
                                           ctypes.cast(bg, ctypes.c_void_p), _lib.ptr(rgb), _lib.ptr(sctx.workspace),
                                           _lib.stream()), "hm_shade_rgb")
    return sctx.crop(rgb), depth, alpha


class _OrdinalDepthLoss(torch.autograd.Function):
    """reference homan/lossutils.py:133-169 for the two layers (object, hand) of homan/homan.py:384-419."""

    @staticmethod
    def forward(ctx, d0, d1, a0, a1, m0, m1, rws):
        d0, d1, a0, a1 = _f32(d0), _f32(d1), _f32(a0), _f32(a1)
        B, S = d0.shape[0], d0.shape[1]
        assert d0.shape == d1.shape == a0.shape == a1.shape == m0.shape == m1.shape == (B, S, S)
        assert m0.dtype == torch.uint8 and m1.dtype == torch.uint8 and m0.is_contiguous() and m1.is_contiguous()
        part = torch.zeros(B * 8, device=d0.device)          # (per-frame records: zero on entry, see the header)
        rec = torch.empty(5, device=d0.device)
        out = torch.empty(1, device=d0.device)
        _lib.check(_lib.lib().hm_ordinal_depth_fwd(
            _lib.ptr(d0), _lib.ptr(d1), _lib.ptr(a0), _lib.ptr(a1), _lib.ptr(m0), _lib.ptr(m1), B, S, _lib.ptr(part),
            _lib.ptr(rec), _lib.ptr(out), _lib.ptr(rws.buf), _lib.stream()), "hm_ordinal_depth_fwd")
        ctx.save_for_backward(d0, d1, a0, a1, m0, m1, rec)
        pairs = rec[:1].clone()             # the normaliser of this pair of layers (a count of frames: no gradient)
        ctx.mark_non_differentiable(pairs)
        return out.reshape(()), pairs

    @staticmethod
    def backward(ctx, g, _g_pairs=None):
        d0, d1, a0, a1, m0, m1, rec = ctx.saved_tensors
        g0, g1 = torch.empty_like(d0), torch.empty_like(d1)
        _lib.check(_lib.lib().hm_ordinal_depth_bwd(
            _lib.ptr(d0), _lib.ptr(d1), _lib.ptr(a0), _lib.ptr(a1), _lib.ptr(m0), _lib.ptr(m1), d0.shape[0], d0.shape[1],
            _lib.ptr(rec), _lib.ptr(_f32(g).reshape(1)), _lib.ptr(g0), _lib.ptr(g1), _lib.stream()), "hm_ordinal_depth_bwd")
        return g0, g1, None, None, None, None, None


def ordinal_depth_loss(d_obj, d_hand, sil_obj, sil_hand, mask_obj, mask_hand, rws):
    """0-d loss.  mask_*: (B,S,S) uint8 instance masks."""
    return _OrdinalDepthLoss.apply(d_obj, d_hand, sil_obj, sil_hand, mask_obj, mask_hand, rws)[0]


def ordinal_depth_loss_layers(depths, sils, masks, rws):
    """The ordinal depth term over n >= 2 layers (reference homan/lossutils.py:133-169: every ordered pair of layers, one
    normaliser for all - the number of (pair, frame) combinations whose two silhouettes meet, a layer with itself included).
    Every unordered pair goes through the two-layer kernels, which normalise by the pair's OWN count; the pair's value times
    (its count / the scene's count) is its share of the scene's loss, and autograd carries the same factor into the gradients.
    depths / sils: n x (B,S,S) float renders, masks: n x (B,S,S) uint8."""
    n = len(depths)
    if n == 2:
        return ordinal_depth_loss(depths[0], depths[1], sils[0], sils[1], masks[0], masks[1], rws)
    with torch.no_grad():
        present = [(s == 1).flatten(1).any(1).sum().float() for s in sils]          # frames in which layer i covers a pixel fully
    terms, counts = [], []
    for a in range(n):
        for b in range(a + 1, n):
            val, pairs = _OrdinalDepthLoss.apply(depths[a], depths[b], sils[a], sils[b], masks[a], masks[b], rws)
            terms.append((val, pairs[0]))
            counts.append(pairs[0] - present[a] - present[b])                       # = 2 x frames in which a and b meet
    total = sum(present) + sum(counts)
    loss = torch.zeros((), device=depths[0].device)
    for val, pairs in terms:
        loss = loss + torch.where(pairs > 0, val * (pairs / total), torch.zeros_like(val))
    return loss


# =============================================================================== shared small state
class ReduceWorkspace:
    """Zero-initialised scratch for the single-launch grid reductions (partials + self-resetting ticket)."""

    def __init__(self, device):
        self.buf = torch.zeros(_lib.lib().hm_reduce_workspace_bytes(), dtype=torch.uint8, device=device)


def _scale_by(unit, g):
    """g (scalar tensor) * unit, in a HIP kernel."""
    out = torch.empty_like(unit)
    g = _f32(g).reshape(1)
    _lib.check(_lib.lib().hm_scale_by(_lib.ptr(unit), _lib.ptr(g), unit.numel(), _lib.ptr(out), _lib.stream()),
               "hm_scale_by")
    return out


# =============================================================================== rigid transform
_RIGID_WS = {}


def _rigid_workspace(N, device):
    """Zero-initialised ticket / partials buffer of hm_rigid_bwd, one per (frames, device, stream)."""
    key = (N, str(device), torch.cuda.current_stream().cuda_stream)
    if key not in _RIGID_WS:
        _RIGID_WS[key] = torch.zeros(_lib.lib().hm_rigid_workspace_bytes(N), dtype=torch.uint8, device=device)
    return _RIGID_WS[key]


class _RigidTransform(torch.autograd.Function):
    """reference homan/utils/geometry.py:9-27 + homan/utils/camera.py:108-139:
    verts = (s * mesh) @ rot6d_to_matrix(rot6d) + t, and the mesh-detached twin (same values; its gradient
    reaches rotation and translation only)."""

    @staticmethod
    def forward(ctx, mesh, rot6d, trans, scale, abs_scale):
        mesh, rot6d, trans, scale = _f32(mesh), _f32(rot6d), _f32(trans), _f32(scale)
        N, V = mesh.shape[0], mesh.shape[1]
        verts = torch.empty_like(mesh)
        _lib.check(_lib.lib().hm_rigid_fwd(_lib.ptr(mesh), _lib.ptr(rot6d), _lib.ptr(trans), _lib.ptr(scale),
                                           int(abs_scale), N, V, None, _lib.ptr(verts), _lib.stream()), "hm_rigid_fwd")
        ctx.save_for_backward(mesh, rot6d, scale)
        ctx.abs_scale = int(abs_scale)
        ctx.trans_shape = trans.shape
        return verts, verts.clone()

    @staticmethod
    def backward(ctx, g_full, g_det):
        mesh, rot6d, scale = ctx.saved_tensors
        N, V = mesh.shape[0], mesh.shape[1]
        need_mesh, need_scale = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        g_full = None if g_full is None else _f32(g_full)
        g_det = None if g_det is None else _f32(g_det)
        g_mesh = torch.empty_like(mesh) if need_mesh else None
        g_rot = torch.empty_like(rot6d)
        g_trans = torch.empty(N, 3, device=mesh.device)
        g_sp = torch.empty(N, device=mesh.device) if need_scale else None
        tp, tw, tn = _lib.terms([(g_full, 1.0)])
        _lib.check(_lib.lib().hm_rigid_bwd(_lib.ptr(mesh), _lib.ptr(rot6d), _lib.ptr(scale), ctx.abs_scale, tp, tw, tn,
                                           _lib.ptr(g_det), None, 0, 0.0, N, V, _lib.ptr(g_mesh), _lib.ptr(g_rot),
                                           _lib.ptr(g_trans), _lib.ptr(g_sp), _lib.ptr(_rigid_workspace(N, mesh.device)),
                                           _lib.stream()), "hm_rigid_bwd")
        g_scale = g_sp.sum().reshape(scale.shape) if need_scale else None
        return g_mesh, g_rot, g_trans.view(ctx.trans_shape), g_scale, None


def rigid_transform(mesh, rot6d, trans, scale, abs_scale=False):
    return _RigidTransform.apply(mesh, rot6d, trans, scale, abs_scale)


# =============================================================================== MANO
class ManoContext:
    """Device copy of the MANO model in the kernel layout (see csrc/mano.hip) + backward workspace."""

    def __init__(self, model_np, device, num_pca_comps=16, flat_hand_mean=False):
        import ctypes
        assert num_pca_comps == 16, "the reference builds ManoModel(pca_comps=16) (homan/homan.py:70)"
        from .mano_assets import kernel_layout
        self.tensors = [torch.from_numpy(a).to(device) for a in kernel_layout(model_np, flat_hand_mean)]
        self.ptrs = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in self.tensors])
        self.device = device
        self._ws = {}

    def workspace(self, B):
        if B not in self._ws:
            self._ws[B] = torch.zeros(_lib.lib().hm_mano_workspace_bytes(B), dtype=torch.uint8, device=self.device)
        return self._ws[B]


class _ManoLBS(torch.autograd.Function):
    """reference homan/manomodel.py:84-151 (forward_pca, right hand) + the `mano` layer + homan/homan.py:356 (+mano_trans)."""

    @staticmethod
    def forward(ctx, pca, rot, betas, trans, mctx):
        pca, rot, betas = _f32(pca), _f32(rot), _f32(betas)
        trans = None if trans is None else _f32(trans)
        B, P = pca.shape
        verts = torch.empty(B, 778, 3, device=pca.device)
        state = torch.empty(_lib.lib().hm_mano_state_bytes(B), dtype=torch.uint8, device=pca.device)
        _lib.check(_lib.lib().hm_mano_fwd(mctx.ptrs, _lib.ptr(pca), P, _lib.ptr(rot), _lib.ptr(betas), _lib.ptr(trans), B,
                                          _lib.ptr(verts), None, None, None, None, None, _lib.ptr(state), _lib.stream()),
                   "hm_mano_fwd")
        ctx.save_for_backward(pca, rot, betas, state)
        ctx.mctx, ctx.has_trans = mctx, trans is not None
        return verts

    @staticmethod
    def backward(ctx, g_verts):
        pca, rot, betas, state = ctx.saved_tensors
        B, P = pca.shape
        g_verts = _f32(g_verts)
        g_pca, g_rot, g_betas = torch.empty_like(pca), torch.empty_like(rot), torch.empty_like(betas)
        g_trans = torch.empty(B, 3, device=pca.device)
        _lib.check(_lib.lib().hm_mano_bwd(ctx.mctx.ptrs, _lib.ptr(pca), P, _lib.ptr(rot), _lib.ptr(betas), B,
                                          _lib.ptr(g_verts), None, 0.0, _lib.ptr(g_pca), _lib.ptr(g_rot), _lib.ptr(g_betas),
                                          _lib.ptr(g_trans), _lib.ptr(state), _lib.ptr(ctx.mctx.workspace(B)),
                                          _lib.stream()), "hm_mano_bwd")
        return g_pca, g_rot, g_betas, (g_trans if ctx.has_trans else None), None


def mano_lbs(pca, rot, betas, trans, mctx):
    return _ManoLBS.apply(pca, rot, betas, trans, mctx)


def mano_joints(pca, rot, betas, trans, mctx):
    """Posed joints (B,16,3), no gradient (reference homan/homan.py:309-339 is off the optimisation path)."""
    pca, rot, betas = _f32(pca.detach()), _f32(rot.detach()), _f32(betas.detach())
    B, P = pca.shape
    verts = torch.empty(B, 778, 3, device=pca.device)
    joints = torch.empty(B, 16, 3, device=pca.device)
    tr = None if trans is None else _f32(trans.detach())
    _lib.check(_lib.lib().hm_mano_fwd(mctx.ptrs, _lib.ptr(pca), P, _lib.ptr(rot), _lib.ptr(betas), _lib.ptr(tr), B,
                                      _lib.ptr(verts), _lib.ptr(joints), None, None, None, None, None, _lib.stream()),
               "hm_mano_fwd")
    return verts, joints


# =============================================================================== vertex losses
class _V2dLoss(torch.autograd.Function):
    """reference homan/losses.py:141-164."""

    @staticmethod
    def forward(ctx, verts, camintr, ref2d, image_size, hand_nb, rws):
        verts = _f32(verts)
        N, V = verts.shape[:2]
        unit = torch.empty_like(verts)
        out = torch.empty(2, device=verts.device)
        _lib.check(_lib.lib().hm_v2d_fwd(_lib.ptr(verts), _lib.ptr(camintr), int(hand_nb), _lib.ptr(ref2d),
                                         float(image_size), N, V, _lib.ptr(unit), _lib.ptr(out), _lib.ptr(rws.buf),
                                         _lib.stream()), "hm_v2d_fwd")
        ctx.save_for_backward(unit)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_loss, _g_metric):
        (unit,) = ctx.saved_tensors
        return _scale_by(unit, g_loss), None, None, None, None, None


def v2d_loss(verts, camintr, ref2d, image_size, hand_nb, rws):
    return _V2dLoss.apply(verts, camintr, ref2d, image_size, hand_nb, rws)


class _SmoothLoss(torch.autograd.Function):
    """reference homan/lossutils.py:18-36 (one term: hand or object)."""

    @staticmethod
    def forward(ctx, verts, hand_nb, rws):
        verts = _f32(verts)
        N, V = verts.shape[:2]
        unit = torch.empty_like(verts)
        out = torch.empty(1, device=verts.device)
        _lib.check(_lib.lib().hm_smooth_fwd(_lib.ptr(verts), N, V, int(hand_nb), _lib.ptr(unit), _lib.ptr(out),
                                            _lib.ptr(rws.buf), _lib.stream()), "hm_smooth_fwd")
        ctx.save_for_backward(unit)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (unit,) = ctx.saved_tensors
        return _scale_by(unit, g), None, None


def smooth_loss(verts, hand_nb, rws):
    return _SmoothLoss.apply(verts, hand_nb, rws)


class _Priors(torch.autograd.Function):
    """reference homan/lossutils.py:39-40 (mean(pca^2)) and :107-109 (scale priors)."""

    @staticmethod
    def forward(ctx, pca, s_obj, m_obj, s_hand, m_hand):
        pca, s_obj, s_hand = _f32(pca), _f32(s_obj), _f32(s_hand)
        g_pca = torch.empty_like(pca)
        g_so, g_sh = torch.empty_like(s_obj), torch.empty_like(s_hand)
        out = torch.empty(3, device=pca.device)
        _lib.check(_lib.lib().hm_priors_fwd(_lib.ptr(pca), pca.numel(), _lib.ptr(s_obj), _lib.ptr(m_obj),
                                            _lib.ptr(s_hand), _lib.ptr(m_hand), _lib.ptr(g_pca), _lib.ptr(g_so),
                                            _lib.ptr(g_sh), _lib.ptr(out), _lib.stream()), "hm_priors_fwd")
        ctx.save_for_backward(g_pca, g_so, g_sh)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        g_pca, g_so, g_sh = ctx.saved_tensors
        return (_scale_by(g_pca, g0) if ctx.needs_input_grad[0] else None,
                _scale_by(g_so, g1) if ctx.needs_input_grad[1] else None, None,
                _scale_by(g_sh, g2) if ctx.needs_input_grad[3] else None, None)


def priors(pca, s_obj, m_obj, s_hand, m_hand):
    return _Priors.apply(pca, s_obj, m_obj, s_hand, m_hand)


class _InterLoss(torch.autograd.Function):
    """reference homan/losses.py:199-242 ('centroid'), gating :98-139; returns the un-normalised sum, shape (1,)."""

    @staticmethod
    def forward(ctx, vh, vo, camintr, expansion, zthresh, rws):
        vh, vo = _f32(vh), _f32(vo)
        B, Vh, Vo = vo.shape[0], vh.shape[1], vo.shape[1]
        assert vh.shape[0] == B, "one hand per frame"
        rec = torch.empty(B, 8, device=vh.device)
        out = torch.empty(1, device=vh.device)
        _lib.check(_lib.lib().hm_inter_fwd(_lib.ptr(vh), _lib.ptr(vo), _lib.ptr(camintr), B, Vh, Vo, float(expansion),
                                           float(zthresh), _lib.ptr(rec), _lib.ptr(out), _lib.ptr(rws.buf),
                                           _lib.stream()), "hm_inter_fwd")
        ctx.save_for_backward(rec)
        ctx.dims = (B, Vh, Vo)
        return out

    @staticmethod
    def backward(ctx, g):
        (rec,) = ctx.saved_tensors
        B, Vh, Vo = ctx.dims
        g = _f32(g).reshape(1)
        gh = torch.empty(B, Vh, 3, device=rec.device) if ctx.needs_input_grad[0] else None
        go = torch.empty(B, Vo, 3, device=rec.device) if ctx.needs_input_grad[1] else None
        if gh is not None or go is not None:
            _lib.check(_lib.lib().hm_inter_bwd(_lib.ptr(rec), _lib.ptr(g), B, Vh, Vo, _lib.ptr(gh), _lib.ptr(go),
                                               _lib.stream()), "hm_inter_bwd")
        return gh, go, None, None, None, None


def inter_loss(vh, vo, camintr, rws, expansion=0.2, zthresh=3.0):
    return _InterLoss.apply(vh, vo, camintr, expansion, zthresh, rws)


def interaction_flags(vh, vo, camintr, rws, expansion=0.2, zthresh=3.0):
    """(B,) bool, no grad: which frames the coarse interaction term applies to (expanded 2-D boxes overlap and the depth ranges
    are closer than `zthresh`; reference homan/losses.py:98-139), from the frame records of hm_inter_fwd."""
    with torch.no_grad():
        vh, vo = _f32(vh.detach()), _f32(vo.detach())
        B = vo.shape[0]
        rec = torch.empty(B, 8, device=vh.device)
        out = torch.empty(1, device=vh.device)
        _lib.check(_lib.lib().hm_inter_fwd(_lib.ptr(vh), _lib.ptr(vo), _lib.ptr(camintr), B, vh.shape[1], vo.shape[1],
                                           float(expansion), float(zthresh), _lib.ptr(rec), _lib.ptr(out), _lib.ptr(rws.buf),
                                           _lib.stream()), "hm_inter_fwd")
        return rec[:, 0] != 0


def nearest_vertices(vh, vo, rws, metric_only=False):
    """hand -> object nearest vertex (no grad): idx (B,Vh) int32, squared distance, and the metric
    max_b min_{i,j} |h_i - o_j| of reference homan/losses.py:225-241.  metric_only: (None, None, metric) - the same exact
    value from the pruned search."""
    vh, vo = _f32(vh.detach()), _f32(vo.detach())
    B, Vh, Vo = vo.shape[0], vh.shape[1], vo.shape[1]
    idx = None if metric_only else torch.empty(B, Vh, dtype=torch.int32, device=vh.device)
    d2 = None if metric_only else torch.empty(B, Vh, device=vh.device)
    metric = torch.empty(1, device=vh.device)
    _lib.check(_lib.lib().hm_nn_fwd(_lib.ptr(vh), _lib.ptr(vo), B, Vh, Vo, _lib.ptr(idx), _lib.ptr(d2), _lib.ptr(metric),
                                    _lib.ptr(rws.buf), _lib.stream()), "hm_nn_fwd")
    return idx, d2, metric


class _ContactLoss(torch.autograd.Function):
    """reference homan/lossutils.py:112-130 -> interactions/contactloss.py:149-309 as executed (appendix B.1)."""

    @staticmethod
    def forward(ctx, vh, vo, nn_idx, thresh, rws):
        vh, vo = _f32(vh), _f32(vo)
        B, Vh, Vo = vo.shape[0], vh.shape[1], vo.shape[1]
        gh, go = torch.empty_like(vh), torch.empty_like(vo)
        out = torch.empty(1, device=vh.device)
        _lib.check(_lib.lib().hm_contact_fwd(_lib.ptr(vh), _lib.ptr(vo), _lib.ptr(nn_idx), B, Vh, Vo, float(thresh),
                                             _lib.ptr(gh), _lib.ptr(go), _lib.ptr(out), _lib.ptr(rws.buf),
                                             _lib.stream()), "hm_contact_fwd")
        ctx.save_for_backward(gh, go)
        return out

    @staticmethod
    def backward(ctx, g):
        gh, go = ctx.saved_tensors
        return (_scale_by(gh, g) if ctx.needs_input_grad[0] else None,
                _scale_by(go, g) if ctx.needs_input_grad[1] else None, None, None, None)


def contact_loss(vh, vo, nn_idx, rws, thresh=0.020):
    return _ContactLoss.apply(vh, vo, nn_idx, thresh, rws)


class CollisionContext:
    def __init__(self, faces_hand_closed, faces_obj, B, Vh, Vo, device):
        self.f0 = torch.as_tensor(np.asarray(faces_hand_closed), dtype=torch.int32).to(device).contiguous()
        self.f1 = faces_obj.to(device=device, dtype=torch.int32).contiguous()
        self.B, self.V0, self.V1 = B, Vh, Vo
        self.ws = torch.zeros(_lib.lib().hm_collision_workspace_bytes(B, Vh, Vo, self.f0.shape[0], self.f1.shape[0]),
                              dtype=torch.uint8, device=device)

    def grid(self, which):
        """clamp(SDF,0) (B,32,32,32) of mesh `which` from the last forward (debug / API completeness)."""
        f = self.f0 if which == 0 else self.f1
        V = self.V0 if which == 0 else self.V1
        phi = torch.empty(self.B, 32, 32, 32, device=self.ws.device)
        _lib.check(_lib.lib().hm_collision_read_grid(_lib.ptr(f), V, f.shape[0], self.B, which, self.V0, self.V1,
                                                     self.f0.shape[0], self.f1.shape[0], _lib.ptr(phi),
                                                     _lib.ptr(self.ws), _lib.stream()),
                   "hm_collision_read_grid")
        return phi


class _CollisionLoss(torch.autograd.Function):
    """reference homan/lossutils.py:43-64 (sdf branch) -> interactions/scenesdf.py:77-148."""

    @staticmethod
    def forward(ctx, vh, vo, cctx, scale_factor):
        vh, vo = _f32(vh), _f32(vo)
        g0, g1 = torch.empty_like(vh), torch.empty_like(vo)
        out = torch.empty(1, device=vh.device)
        _lib.check(_lib.lib().hm_collision_fwd(_lib.ptr(vh), _lib.ptr(cctx.f0), cctx.V0, cctx.f0.shape[0], _lib.ptr(vo),
                                               _lib.ptr(cctx.f1), cctx.V1, cctx.f1.shape[0], cctx.B, float(scale_factor),
                                               _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(out), _lib.ptr(cctx.ws),
                                               _lib.stream()), "hm_collision_fwd")
        ctx.save_for_backward(g0, g1)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        g0, g1 = ctx.saved_tensors
        return (_scale_by(g0, g) if ctx.needs_input_grad[0] else None,
                _scale_by(g1, g) if ctx.needs_input_grad[1] else None, None, None)


def collision_loss(vh, vo, cctx, scale_factor=0.2):
    return _CollisionLoss.apply(vh, vo, cctx, scale_factor)


def collision_dist_values(vh, vo, cctx, scale_factor=0.2):
    """`sdf_meta["dist_values"]` of reference homan/interactions/scenesdf.py:141-146 for the two-mesh scene
    [hand, object]: {(1, 0): (B,Vh) depth of the hand vertices inside the object, (0, 1): (B,Vo) the reverse}, world
    units, no gradient.  Runs the SDF forward on (vh, vo) and samples its grids (csrc/sdf.hip)."""
    with torch.no_grad():
        vh, vo = _f32(vh.detach()), _f32(vo.detach())
        collision_loss(vh, vo, cctx, scale_factor)
        dv0 = torch.empty(vh.shape[:2], device=vh.device)
        dv1 = torch.empty(vo.shape[:2], device=vo.device)
        _lib.check(_lib.lib().hm_collision_dist_values(_lib.ptr(vh), cctx.V0, _lib.ptr(vo), cctx.V1, cctx.f0.shape[0],
                                                       cctx.f1.shape[0], cctx.B, _lib.ptr(dv0), _lib.ptr(dv1),
                                                       _lib.ptr(cctx.ws), _lib.stream()), "hm_collision_dist_values")
    return {(1, 0): dv0, (0, 1): dv1}
