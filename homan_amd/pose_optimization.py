"""Object-pose initialisation from an instance mask on the HIP rasteriser.

Mirrors reference homan/pose_optimization.py: `PoseOptimizer` (:37-160, same constructor keywords minus `textures`,
same parameter names `rotations` / `translations`, same `forward() -> (loss_dict, iou, image)`) and
`find_optimal_pose` (:219-383, same arguments and returned module; the debug plots are not provided).  The render leaf
nr.Renderer(image_size, anti_aliasing=False) is `ops.silhouette_render_noaa` (csrc/raster_fwd.hip, hm_sil_fwd `alpha_full`
+ hm_sil_bwd mode 3); everything else is host logic in torch, as in the reference.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
from scipy.ndimage import distance_transform_edt

from . import constants, ops
from . import lib as _lib
from .homan import matrix_to_rot6d

NMR_FAR = 100.0


def compute_random_rotations(B=10, device="cpu"):
    """reference homan/utils/geometry.py:89-134 (upright=False): uniform rotations (Arvo 1992)."""
    x1, x2, x3 = torch.split(torch.rand(3 * B), B)
    tau = 2 * math.pi
    zeros, ones = torch.zeros_like(x1), torch.ones_like(x1)
    R = torch.stack((torch.stack((torch.cos(tau * x1), torch.sin(tau * x1), zeros), 1),
                     torch.stack((-torch.sin(tau * x1), torch.cos(tau * x1), zeros), 1),
                     torch.stack((zeros, zeros, ones), 1)), 1)
    v = torch.stack((torch.cos(tau * x2) * torch.sqrt(x3), torch.sin(tau * x2) * torch.sqrt(x3), torch.sqrt(1 - x3)), 1)
    H = torch.eye(3).repeat(B, 1, 1) - 2 * v.unsqueeze(2) * v.unsqueeze(1)
    return (-torch.matmul(H, R)).to(device)


def get_K_crop_resize(K, boxes, crop_resize):
    """libyana.lib3d.kcrop.get_K_crop_resize (reference homan/pose_optimization.py:246-248): intrinsics of the crop
    `boxes` (xyxy, pixels) resized to `crop_resize`; pixel-centre convention of the cosypose routine it derives from."""
    K = K.float().clone()
    boxes = boxes.float()
    crop_resize = torch.as_tensor(crop_resize, dtype=torch.float32)
    final_w, final_h = crop_resize.max(), crop_resize.min()
    crop_w, crop_h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    crop_cj, crop_ci = (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2
    cx = K[:, 0, 2] + (crop_w - 1) / 2 - crop_cj
    cy = K[:, 1, 2] + (crop_h - 1) / 2 - crop_ci
    scale_x, scale_y = final_w / crop_w, final_h / crop_h
    new_K = K.clone()
    new_K[:, 0, 0] = scale_x * K[:, 0, 0]
    new_K[:, 1, 1] = scale_y * K[:, 1, 1]
    new_K[:, 0, 2] = (final_w - 1) / 2 + scale_x * (cx - (crop_w - 1) / 2)
    new_K[:, 1, 2] = (final_h - 1) / 2 + scale_y * (cy - (crop_h - 1) / 2)
    return new_K


def _centre_and_diagonal(lo, hi):
    """axis-aligned 2-D boxes given by their corners (n,2) -> centres (n,2), diagonal lengths (n,)"""
    return (lo + hi) / 2, (hi - lo).norm(dim=-1)


def TCO_init_from_boxes_zup_autodepth(boxes_2d, model_points_3d, K):
    """Translation that makes the projected bounding box of the points match an xywh target box (counterpart of reference
    homan/lib3d/optitrans.py:83-127, same name so that callers of the reference find it): start on the ray through the
    box centre at unit depth, then ten fixed-point rounds - rescale the depth by the ratio of the box diagonals, shift
    x / y by the back-projected offset of the box centres."""
    pts = torch.as_tensor(model_points_3d).float()
    n, dev = pts.shape[0], pts.device
    cam = torch.as_tensor(K, dtype=torch.float32).to(dev).reshape(-1, 3, 3).expand(n, 3, 3)
    xywh = torch.as_tensor(boxes_2d, dtype=torch.float32).to(dev).reshape(-1, 4).expand(n, 4)
    target_centre, target_diag = _centre_and_diagonal(xywh[:, :2], xywh[:, :2] + xywh[:, 2:])
    focal = torch.stack([cam[:, 0, 0], cam[:, 1, 1]], -1)
    principal = cam[:, :2, 2]
    depth = torch.ones(n, 1, device=dev)
    xy = (target_centre - principal) * depth / focal
    for _ in range(10):
        placed = pts + torch.cat([xy, depth], 1)[:, None]
        hom = placed @ cam.transpose(1, 2)
        uv = hom[..., :2] / hom[..., 2:]
        centre, diag = _centre_and_diagonal(uv.amin(1), uv.amax(1))
        depth = depth * (diag / target_diag)[:, None]           # (= depth + depth * (ratio - 1))
        xy = xy + (target_centre - centre) * depth / focal
    return torch.cat([xy, depth], 1)


class PoseOptimizer(nn.Module):
    """reference homan/pose_optimization.py:37-160 (occlusion-aware silhouette loss + one-way edge chamfer + off-screen
    penalty over `num_initializations` candidate poses of one mesh against one instance mask)."""

    def __init__(self, ref_image, vertices, faces, rotation_init, translation_init, num_initializations=1, kernel_size=7,
                 K=None, power=0.25, lw_chamfer=0, textures=None, _shared=None):
        assert ref_image.shape[0] == ref_image.shape[1], "Must be square."
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("homan_amd.pose_optimization needs an MI355X (ROCm) device; there is no CPU path")
        dev = torch.device("cuda")
        size = int(ref_image.shape[0])
        if size % 2:
            raise NotImplementedError(f"pose initialisation renders on a grid of 2x2-sample pixels: even mask sizes only "
                                      f"(the reference's REND_SIZE is 256), got {size}")
        # (sizes off the kernels' 64-sample tile grid are rendered on the next one and cropped: ops.SilhouetteContext)
        # The reference builds n-fold copies of the mask, its keep mask and the edge distance transform on the host and uploads
        # ~400 MB per fit (two thirds of a whole 50-step fit here).  The kernels read ONE copy of each (`_keep1`, `_ref1`);
        # `image_ref` / `keep_mask` / `edt_ref_edge` are still there under the reference's names, as broadcast views.
        n = self._n = num_initializations
        self._power, self._kernel_size = power, kernel_size
        # `_shared`: another PoseOptimizer of the same mesh, candidate count and mask size (the resident fitter's): the mesh
        # copies and the rasteriser's context - which hold nothing of a fit between two calls - are used, not rebuilt
        if _shared is not None:
            self.register_buffer("vertices", _shared.vertices, persistent=False)
            self.register_buffer("faces", _shared.faces, persistent=False)
        else:
            tile = lambda t: t.to(dev).repeat(n, *([1] * (t.dim())))          # (…) -> (n, …)
            self.register_buffer("vertices", tile(torch.as_tensor(vertices).float().reshape(-1, 3)))
            self.register_buffer("faces", tile(torch.as_tensor(faces).reshape(-1, 3)))
        # instance mask convention (:66-74): -1 = occluded (not compared), 0 = background, 1 = object
        mask = torch.as_tensor(np.asarray(ref_image)).float().to(dev)
        self._ref1, self._keep1 = (mask > 0).float().contiguous(), (mask >= 0).float().contiguous()
        self.pool = torch.nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=(kernel_size // 2))
        # candidate poses: 6-D rotations (n,3,2) and translations (n,1,3); a single translation serves every candidate
        rot0, trans0 = torch.as_tensor(rotation_init).float(), torch.as_tensor(translation_init).float()
        if trans0.shape[0] != rot0.shape[0]:
            trans0 = trans0.repeat(n, 1, 1)
        self.rotations = nn.Parameter(rot0.clone().to(dev), requires_grad=True)
        self.translations = nn.Parameter(trans0.clone().to(dev), requires_grad=True)
        self._edt1 = None           # one-way chamfer term's distance transform (:76-80): built when first read (weight 0 upstream)
        if K is None:
            K = torch.tensor([[[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]]], dtype=torch.float32)
        self.register_buffer("K", torch.as_tensor(K).float().reshape(-1, 3, 3)[:1].clone())
        self.image_size, self.lw_chamfer = size, lw_chamfer
        self.to(dev)
        self._K_all = self.K.repeat(n, 1, 1).contiguous()
        if _shared is not None:
            self._one, self._sil_ctx = _shared._one, _shared._sil_ctx
        else:
            self._one = torch.ones(1, device=dev)
            self._sil_ctx = ops.SilhouetteContext(self.faces, self.vertices.shape[1], n, size // 2, dev)
            # the masked L2 of :138-143 is an unnormalised sum of squares: per-sample gradients are O(1), pseudo-gradient terms
            # up to 2 / eps, per-frame sums up to ~1e7 - the order-independent sums of the backward run on the grid 2^-24 (exact
            # up to 5e8, resolution 6e-8) instead of the joint fit's 2^-44
            self._sil_ctx.sum_log2q = -24

    # the reference's per-candidate buffers, as views of the single copies
    @property
    def image_ref(self):
        return self._ref1[None].expand(self._n, -1, -1)

    @property
    def keep_mask(self):
        return self._keep1[None].expand(self._n, -1, -1)

    @property
    def edt_ref_edge(self):
        if self._edt1 is None:
            band = self.compute_edges(self._ref1[None, None])[0, 0].cpu().numpy() > 0
            self._edt1 = torch.from_numpy(distance_transform_edt(~band) ** (self._power * 2)).float().to(self._ref1.device)
        return self._edt1[None].expand(self._n, -1, -1)

    def apply_transformation(self):
        """:98-103: vertices @ rot6d_to_matrix(rotations) + translations (csrc/geometry.hip, unit scale)."""
        return ops.rigid_transform(self.vertices, self.rotations, self.translations, self._one, False)[0]

    def compute_offscreen_loss(self, verts):
        """:112-135 with nr.projection(K, R=I, t=0, orig_size=1) written out: on-screen = NDC xy in [-1,1], 0 < z < far."""
        x, y, z = verts[:, :, 0], verts[:, :, 1], verts[:, :, 2]
        xn, yn = x / (z + 1e-9), y / (z + 1e-9)
        k = self.K[0]
        u = k[0, 0] * xn + k[0, 1] * yn + k[0, 2]
        v = 1.0 - (k[1, 0] * xn + k[1, 1] * yn + k[1, 2])
        ndc = torch.stack([2 * (u - 0.5), 2 * (v - 0.5)], -1)            # (n,V,2), on screen iff inside [-1,1]^2
        relu = torch.nn.functional.relu
        # hinge on each of the six clipping planes, summed per candidate: right/bottom, left/top, behind the camera, beyond far
        return (relu(ndc - 1).sum(dim=(1, 2)) + relu(-1 - ndc).sum(dim=(1, 2)) + relu(-z).sum(dim=1) +
                relu(z - NMR_FAR).sum(dim=1))

    def compute_edges(self, silhouette):
        return self.pool(silhouette) - silhouette

    def forward(self):
        verts = self.apply_transformation()
        loss_dict = {}
        if self.lw_chamfer == 0:
            # render + keep-mask + per-pose L2 + IoU in the rasteriser's epilogue (hm_sil_fwd with a per-sample mask)
            mask_loss, iou, alpha = ops.masked_silhouette_l2_noaa(verts, self._K_all, self._keep1, self._ref1,
                                                                  self._sil_ctx, 1.0)
            loss_dict["mask"] = mask_loss
            iou = iou.detach()
            image = self.keep_mask * alpha
            # the reference still max-pools 500 images forward and backward for a term it multiplies by zero (a third of
            # its step); 0 * finite = 0 with a zero gradient, so the value and the gradients are the same without it
            loss_dict["chamfer"] = torch.zeros_like(mask_loss)
        else:
            image = self.keep_mask * ops.silhouette_render_noaa(verts, self._K_all, self._sil_ctx, 1.0)
            loss_dict["mask"] = torch.sum((image - self.image_ref) ** 2, dim=(1, 2))
            with torch.no_grad():
                inter = (image * self.image_ref).sum((1, 2))
                union = (image + self.image_ref).clamp(0, 1).sum((1, 2))
                iou = inter / (union + 1e-6)            # libyana batch_mask_iou
            loss_dict["chamfer"] = self.lw_chamfer * torch.sum(self.compute_edges(image) * self.edt_ref_edge, dim=(1, 2))
        loss_dict["offscreen"] = 100000 * self.compute_offscreen_loss(verts)
        return loss_dict, iou, image


def _graph_loop(model, lr, num_iterations):
    """`num_iterations` steps of reference pose_optimization.py:330-357 replayed from one captured hipGraph.  The
    best-ever bookkeeping keeps the reference's order (the pose is copied AFTER the optimiser step that followed the
    evaluation, :348-353) and its strict `<`."""
    params = [model.rotations, model.translations]
    optimizer = torch.optim.Adam(params, lr=lr, capturable=True)
    dev = model.rotations.device
    best_loss = torch.full((), float("inf"), device=dev)
    best_rot, best_trans = torch.zeros_like(model.rotations[0]), torch.zeros_like(model.translations[0])
    losses_out = torch.zeros(model.rotations.shape[0], device=dev)

    def step():
        loss_dict, _iou, _sil = model()
        losses = sum(loss_dict.values())
        losses.sum().backward()
        optimizer.step()
        with torch.no_grad():
            lmin, ind = losses.min(0)
            better = lmin < best_loss
            best_loss.copy_(torch.where(better, lmin, best_loss))
            sel = ind.reshape(1)             # device-side gather: indexing with a 0-d tensor would sync on its value
            best_rot.copy_(torch.where(better, model.rotations.index_select(0, sel)[0], best_rot))
            best_trans.copy_(torch.where(better, model.translations.index_select(0, sel)[0], best_trans))
            losses_out.copy_(losses)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # two un-captured steps: lazy initialisation of the optimiser state and of the allocator pools; they ARE steps
        for p in params:
            p.grad = torch.zeros_like(p)
        done = 0
        for _ in range(min(2, num_iterations)):
            for p in params:
                p.grad.zero_()
            step()
            done += 1
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if done < num_iterations:
        graph = _lib.new_graph()
        for p in params:
            p.grad.zero_()
        with torch.cuda.graph(graph, pool=_lib.autograd_pool()):
            step()
            for p in params:
                p.grad.zero_()
        for _ in range(num_iterations - done):       # (capturing records the step, it does not run it)
            graph.replay()
    torch.cuda.synchronize()
    return losses_out, best_rot.clone(), best_trans.clone()


class _FusedPoseLoop:
    """The step of reference pose_optimization.py:330-357 as a fixed sequence of C-ABI launches, no autograd tape, captured
    once in a hipGraph and replayed: rigid transform of the n candidates, off-screen penalty (value + vertex gradients in
    one launch, hm_offscreen_fwd), no-anti-aliasing raster with the masked L2 + IoU fused per sample, edge sweeps, pose
    gradients with the silhouette gather inside (hm_rigid_bwd_sil), the fused multi-tensor Adam - what the eager loop spends
    on torch's element-wise kernels (a third of its step) is gone.  The chamfer term is multiplied by its weight 0 at the
    reference's only call site and is not evaluated (PoseOptimizer.forward does the same).  Best-ever bookkeeping as in
    `_graph_loop`: the pose is copied AFTER the optimiser step that followed the evaluation (:348-353), strict `<`.

    The loop is BOUND to one PoseOptimizer: the graph reads its `rotations` / `translations` / `_keep1` / `_ref1` / `K` where
    they lie.  `restart()` makes it the loop of a new fit whose data the caller copied INTO those tensors (PoseFitter)."""

    def __init__(self, model, lr, log_steps=0):
        """log_steps > 0: the loop of ONE GROUP of a fit's candidates (PoseFitter): no best-ever state here - every step leaves its
        record {minimum, candidate, NaN flag, that candidate's pose} in `self.log` (log_steps, 16) and the fitter applies the
        best-ever rule over all groups afterwards (hm_pose_keep_best_log)."""
        from .jointopt import HmAdam
        assert model.lw_chamfer == 0, "the fused loop covers the reference's configuration (lw_chamfer = 0)"
        self.model, self.lr = model, lr
        L, P, ck = _lib.lib(), _lib.ptr, _lib.check
        sctx, dev = model._sil_ctx, model.rotations.device
        n, V, F, S = sctx.B, sctx.V, sctx.F, sctx.S
        f = lambda *shape: torch.zeros(*shape, device=dev)
        verts, g_off, off = f(n, V, 3), f(n, V, 3), f(n)
        pooled, frame = f(n, S, S), f(n, 2)
        # (views of the model's own tensors when the mask size lies on the kernels' tile grid, copies otherwise: refresh())
        self.keep, self.ref = sctx.pad_samples(model._keep1).contiguous(), sctx.pad_samples(model._ref1).contiguous()
        self.K_all, self.K_one, eps = sctx.K_eff(model._K_all).contiguous(), model.K[0].contiguous(), sctx.eps()
        keep, ref, K_all, K_one = self.keep, self.ref, self.K_all, self.K_one
        ones = torch.ones(n, device=dev)
        rws = torch.zeros(L.hm_rigid_workspace_bytes(n), dtype=torch.uint8, device=dev)
        params = self.params = [model.rotations, model.translations]
        for p in params:
            p.grad = torch.zeros_like(p)
        opt = self.opt = HmAdam([{"params": params, "lr": lr}])
        tp, tw, tn = _lib.terms([(g_off, 1.0)])
        self.best_loss = best_loss = torch.full((1,), float("inf"), device=dev)
        self.best_rot, self.best_trans = torch.zeros_like(model.rotations[0]), torch.zeros_like(model.translations[0])
        best_rot, best_trans = self.best_rot, self.best_trans
        self.losses_out = losses_out = f(n)
        self.log_steps = int(log_steps)
        self.log = log = f(self.log_steps, 16) if self.log_steps > 0 else None
        self._keepalive = (verts, g_off, off, pooled, frame, ones, rws, tp, tw)

        def step():
            st = _lib.stream()
            ck(L.hm_rigid_fwd(P(model.vertices), P(model.rotations), P(model.translations), P(model._one), 0, n, V, None, P(verts),
                              st), "hm_rigid_fwd")
            ck(L.hm_offscreen_fwd(P(verts), P(K_one), n, V, NMR_FAR, 100000.0, P(off), P(g_off), st), "hm_offscreen_fwd")
            # (mask_shared = 1 | 2: one binary mask for every candidate, no per-sample outputs - see include/homan_amd.h)
            ck(L.hm_sil_fwd(P(verts), P(sctx.faces), 0, P(K_all), n, V, F, S, 1.0, ops.NMR_NEAR, ops.NMR_FAR, P(keep), P(ref), None,
                            P(pooled), None, P(sctx.work_order), None, None, 3, None, None, None, 0, 0, P(sctx.workspace), st),
               "hm_sil_fwd")
            ck(L.hm_sil_reduce(n, V, F, S, None, None, P(frame), P(sctx.workspace), st), "hm_sil_reduce")
            ck(L.hm_sil_bwd(P(verts), P(K_all), n, V, F, S, 1.0, eps, 5, P(ones), None, None, P(sctx.adj_off), P(sctx.adj_items),
                            P(sctx.face_order), None, None, P(sctx.workspace), sctx.sum_log2q, st), "hm_sil_bwd")
            ck(L.hm_rigid_bwd_sil(P(model.vertices), P(model.rotations), P(model._one), 0, tp, tw, tn,
                                  L.hm_sil_parts(P(sctx.workspace), n, V, F, S), P(sctx.adj_off), P(sctx.adj_items), P(verts),
                                  P(K_all), 1.0, F, n, V, P(model.rotations.grad), P(model.translations.grad), None, P(rws),
                                  sctx.sum_log2q, st),
               "hm_rigid_bwd_sil")
            opt.step(zero_grad=False)
            # mask + (chamfer = 0) + offscreen, the order of sum(loss_dict.values()); best-ever bookkeeping in the same launch
            if log is not None:
                ck(L.hm_pose_keep_best_log(P(frame), 2, P(off), n, P(model.rotations), P(model.translations), P(opt.step_t),
                                           self.log_steps, P(log), P(losses_out), st), "hm_pose_keep_best_log")
            else:
                ck(L.hm_pose_keep_best(P(frame), 2, P(off), n, P(model.rotations), P(model.translations), P(best_loss), P(best_rot),
                                       P(best_trans), P(losses_out), st), "hm_pose_keep_best")

        self._step = step
        self.graph = None

    def restart(self):
        """a new fit in the bound model's tensors: derived inputs refreshed, optimiser and best-ever state as new"""
        m, sctx = self.model, self.model._sil_ctx
        for dst, src in ((self.keep, sctx.pad_samples(m._keep1)), (self.ref, sctx.pad_samples(m._ref1)),
                         (self.K_all, sctx.K_eff(m._K_all)), (self.K_one, m.K[0])):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        for mm, vv in self.opt.state:
            mm.zero_()
            vv.zero_()
        self.opt.step_t.zero_()
        for p in self.params:
            p.grad.zero_()
        self.best_loss.fill_(float("inf"))
        sctx.invalidate_outputs()

    def run(self, num_iterations):
        """`num_iterations` steps.  The first two steps of a loop's life run un-captured (lazy initialisation of the library)
        - they ARE steps, same launches - then one step is captured and every further step, of this and of later fits, is a
        replay.  -> (final losses (n,), best-ever rotation, best-ever translation)"""
        done = self.prepare(num_iterations)
        for _ in range(num_iterations - done):
            self.graph.replay()
        return self.losses_out, self.best_rot.clone(), self.best_trans.clone()

    def prepare(self, num_iterations):
        """the un-captured first steps of the loop's life and the capture (see run) -> steps done (0 once the graph exists)"""
        done = 0
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(min(2, num_iterations)):
                    self._step()
                    done += 1
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if done < num_iterations:
                self.graph = _lib.new_graph()
                with torch.cuda.graph(self.graph):
                    self._step()
        return done

    def stamped_replays(self, stamp_reps):
        """(bench.py) `stamp_reps` MORE replays with the heavy silhouette kernels stamping the device wall clock
        (hm_sil_timestamps) -> their average durations in microseconds (raster, lines, sweep)"""
        import ctypes
        L, P, ck = _lib.lib(), _lib.ptr, _lib.check
        sctx = self.model._sil_ctx
        ws, dims = P(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
        saved = torch.zeros(stamp_reps, L.hm_sil_timestamps_bytes(*dims) // 8, dtype=torch.int64, device=sctx.workspace.device)
        for i in range(stamp_reps):
            ck(L.hm_sil_timestamps(ws, *dims, 1, _lib.stream()), "hm_sil_timestamps")
            self.graph.replay()
            ck(L.hm_sil_timestamps_save(ws, *dims, saved[i].data_ptr(), _lib.stream()), "hm_sil_timestamps_save")
        ck(L.hm_sil_timestamps(ws, *dims, 0, _lib.stream()), "hm_sil_timestamps")
        us3, acc = (ctypes.c_float * 3)(), [0.0, 0.0, 0.0]
        for i in range(stamp_reps):
            ck(L.hm_sil_timestamps_read(None, *dims, saved[i].data_ptr(), ctypes.cast(us3, ctypes.c_void_p), _lib.stream()),
               "hm_sil_timestamps_read")
            acc = [a + u for a, u in zip(acc, us3)]
        torch.cuda.synchronize()
        return dict(zip(("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep"), (a / stamp_reps for a in acc)))

    def release(self):
        for p in self.params:
            p.grad = None


def _fused_loop(model, lr, num_iterations, stamp_reps=0):
    """One fit of `model` by a fused loop of its own (`_FusedPoseLoop`).  stamp_reps > 0 (bench.py): that many MORE replays of
    the same graph with the kernels' timestamps on; their average durations are returned as a 4th element."""
    loop = _FusedPoseLoop(model, lr)
    out = loop.run(num_iterations)
    torch.cuda.synchronize()
    stamps = loop.stamped_replays(stamp_reps) if (stamp_reps > 0 and loop.graph is not None) else None
    loop.release()
    return out + (stamps,) if stamp_reps > 0 else out


class PoseFitter:
    """A RESIDENT pose initialiser for one mesh, candidate count and mask size: the mesh copies, the rasteriser's context, the
    loop's buffers and its captured hipGraph are built once; every further fit copies its mask, intrinsics and starting poses
    into place and replays.  `find_optimal_pose` keeps one per (mesh, n, size, lr) - the per-frame fits of `find_optimal_poses`
    (reference homan/pose_optimization.py:386-488: one fit per frame of a clip, same mesh) pay the construction once.

    From 48 candidates on the fitter walks them as two, from 96 on as THREE GROUPS (HOMAN_POSE_PARTS): a resident loop each - own
    mesh copies, rasteriser context, Adam state, hipGraph - replayed side by side on streams of their own.  The candidates of a fit are independent but
    for the best-ever bookkeeping, and a step is one dependent chain of launches: with several chains in flight the small launches
    and tails of one group (a seventh of a step) run under the other groups' rasteriser and sweeps (bench.py --pose-init 500:
    476 k -> 565 k pose-steps/s with three groups; the probe behind it: tools/poseinit_split_probe.py).  Every group's loop leaves one record per step
    (hm_pose_keep_best_log); the rule of reference :340-353 - no update in a step with a NaN loss anywhere, first minimum over
    all candidates, strict `<` - is applied over the groups' records after the last step: the same champion, the same poses and
    losses as one loop over all candidates, bit for bit (tests/test_poseinit.py)."""

    def __init__(self, vertices, faces, num_initializations, size, lr):
        n = num_initializations
        # (same-box bench.py --pose-init N, pose-steps/s with 1 / 2 / 3 / 4 groups: N = 128: 312 k / 354 k / 371 k; 250: 410 k / 420 k /
        #  467 k; 500: 476 k / 533 k / 565 k / 517 k; 2000: 502 k / 574 k / 585 k)
        parts = int(os.environ.get("HOMAN_POSE_PARTS", "0")) or (3 if n >= 96 else 2 if n >= 48 else 1)
        self.parts = parts = max(1, min(parts, n))
        self.cuts = [n * i // parts for i in range(parts + 1)]
        self.shells, self.loops = [], []
        for lo, hi in zip(self.cuts[:-1], self.cuts[1:]):
            rot0 = matrix_to_rot6d(torch.eye(3)[None].repeat(hi - lo, 1, 1))
            shell = PoseOptimizer(ref_image=np.zeros((size, size), np.float32), vertices=vertices, faces=faces, rotation_init=rot0,
                                  translation_init=torch.tensor([[[0.0, 0.0, 1.0]]]), num_initializations=hi - lo,
                                  K=torch.tensor([[[1.0, 0, 0.5], [0, 1.0, 0.5], [0, 0, 1]]]))
            self.shells.append(shell)
        self.lr = lr
        self.shell = self.shells[0]                # (one group: the loop's own best-ever state, as before)
        self.loop = _FusedPoseLoop(self.shell, lr) if parts == 1 else None
        self.streams = [torch.cuda.Stream() for _ in range(parts)] if parts > 1 else []
        self.n, self.size = n, size
        self.fits = 0
        if parts > 1:
            # the (n,)-sized views a fit's result module shares (PoseOptimizer(_shared=...)): the first group's mesh copies tiled
            # to n candidates would be a second copy of the mesh per candidate - the result module gets a shell of its own
            # instead, without a rasteriser workspace until someone renders with it
            self._result_shell = None

    def _loops_for(self, num_iterations):
        if not self.loops:
            self.loops = [_FusedPoseLoop(sh, self.lr, log_steps=max(int(num_iterations), 64)) for sh in self.shells]
        elif self.loops[0].log_steps < num_iterations:         # (a longer fit than any before: larger logs, new graphs)
            for lp in self.loops:
                lp.release()
            self.loops = [_FusedPoseLoop(sh, self.lr, log_steps=int(num_iterations)) for sh in self.shells]
        return self.loops

    def fit(self, mask, rotation_init, translation_init, K, num_iterations):
        """-> (PoseOptimizer holding the fitted candidates in their original order, final losses, champion rotation, champion
        translation); the returned module shares this fitter's mesh copies and rasteriser context, nothing of the fit."""
        if self.parts == 1:
            return self._fit_one(mask, rotation_init, translation_init, K, num_iterations)
        if self._result_shell is None:
            sh0 = self.shells[0]
            self._result_shell = PoseOptimizer(ref_image=np.zeros((self.size, self.size), np.float32), vertices=sh0.vertices[0],
                                               faces=sh0.faces[0], rotation_init=matrix_to_rot6d(torch.eye(3)[None].repeat(self.n, 1, 1)),
                                               translation_init=torch.tensor([[[0.0, 0.0, 1.0]]]), num_initializations=self.n,
                                               K=torch.tensor([[[1.0, 0, 0.5], [0, 1.0, 0.5], [0, 0, 1]]]))
        result = PoseOptimizer(ref_image=mask, vertices=None, faces=None, rotation_init=rotation_init,
                               translation_init=translation_init, num_initializations=self.n, K=K, _shared=self._result_shell)
        loops = self._loops_for(num_iterations)
        with torch.no_grad():
            for (lo, hi), sh in zip(zip(self.cuts[:-1], self.cuts[1:]), self.shells):
                sh._keep1.copy_(result._keep1)
                sh._ref1.copy_(result._ref1)
                sh.K.copy_(result.K)
                sh._K_all.copy_(result._K_all[lo:hi])
                sh.rotations.copy_(result.rotations[lo:hi])
                sh.translations.copy_(result.translations[lo:hi])
        done = []
        for lp in loops:
            lp.restart()
            done.append(lp.prepare(num_iterations))            # (a loop's first steps and its capture: one loop after the other)
        main = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(main)
        for k in range(max(num_iterations - d for d in done)):
            for lp, st, d in zip(loops, self.streams, done):
                if k < num_iterations - d:
                    with torch.cuda.stream(st):
                        lp.graph.replay()
        for st in self.streams:
            main.wait_stream(st)
        # the best-ever rule over the groups' per-step records (k_pose_keep_best's own: NaN anywhere -> no update; the lowest
        # value, the lowest candidate among equals; strict `<` against the best so far)
        logs = torch.stack([lp.log[:num_iterations] for lp in loops]).cpu().numpy()        # (parts, T, 16)
        best, champ = np.float32(np.inf), None
        for t in range(num_iterations):
            rows = logs[:, t]
            if (rows[:, 2] != 0).any():
                continue
            idx = rows[:, 1].copy().view(np.int32)
            cand = [(rows[g, 0], self.cuts[g] + int(idx[g]), g) for g in range(self.parts) if idx[g] >= 0]
            if not cand:
                continue
            v, _, g = min(cand, key=lambda c: (c[0], c[1]))
            if v < best:
                best, champ = v, rows[g, 3:12].copy()
        dev = result.rotations.device
        if champ is None:
            champ_rot, champ_trans = torch.zeros(3, 2, device=dev), torch.zeros(1, 3, device=dev)
        else:
            champ_rot = torch.from_numpy(champ[:6]).reshape(3, 2).to(dev)
            champ_trans = torch.from_numpy(champ[6:9]).reshape(1, 3).to(dev)
        with torch.no_grad():
            for (lo, hi), sh in zip(zip(self.cuts[:-1], self.cuts[1:]), self.shells):
                result.rotations[lo:hi].copy_(sh.rotations)
                result.translations[lo:hi].copy_(sh.translations)
        losses = torch.cat([lp.losses_out for lp in loops])
        self.fits += 1
        return result, losses, champ_rot, champ_trans

    def _fit_one(self, mask, rotation_init, translation_init, K, num_iterations):
        sh = self.shell
        result = PoseOptimizer(ref_image=mask, vertices=None, faces=None, rotation_init=rotation_init,
                               translation_init=translation_init, num_initializations=self.n, K=K, _shared=sh)
        with torch.no_grad():
            sh._keep1.copy_(result._keep1)
            sh._ref1.copy_(result._ref1)
            sh.K.copy_(result.K)
            sh._K_all.copy_(result._K_all)
            sh.rotations.copy_(result.rotations)
            sh.translations.copy_(result.translations)
        self.loop.restart()
        losses, champ_rot, champ_trans = self.loop.run(num_iterations)
        with torch.no_grad():
            result.rotations.copy_(sh.rotations)
            result.translations.copy_(sh.translations)
        self.fits += 1
        return result, losses.clone(), champ_rot, champ_trans


from collections import OrderedDict

_FITTERS = OrderedDict()        # least recently used first


def _fitters_max():
    """Resident fitters kept (each holds the rasteriser's workspace of its n candidates: ~1 GB at 500 x 256^2).  Default 2 - the
    mesh of the clip being initialised and the one before -; HOMAN_POSE_FITTERS_MAX overrides, release_pose_fitters() frees."""
    import os
    return max(1, int(os.environ.get("HOMAN_POSE_FITTERS_MAX", "2")))


def release_pose_fitters():
    """Drops every resident PoseFitter (device buffers, workspaces, captured graphs' buffers) - call it between the pose
    initialisation of a dataset walk and the joint fits when the memory is wanted back.  PoseOptimizers returned by earlier
    fits stay valid (they hold references to what they share)."""
    _FITTERS.clear()


def _digest(t):
    """digest of a mesh tensor's BYTES, taken on every call (one pass over a few tens of KB next to a 50 ms fit).  A cache keyed
    on (address, version, shape) is wrong: the allocator hands the block of a freed mesh to the next mesh of the same shape
    (version 0 again), which then found the previous mesh's digest - and with it the previous mesh's resident fitter."""
    return hash((tuple(t.shape), str(t.dtype), t.detach().cpu().contiguous().numpy().tobytes()))


def _resident_fitter(vertices, faces, n, size, lr, mesh_key=None):
    import os
    if os.environ.get("HOMAN_POSE_FITTER", "1") == "0":
        return None
    mesh_key = mesh_key or (_digest(vertices), _digest(faces))
    key = (*mesh_key, tuple(vertices.shape), tuple(faces.shape), int(n), int(size), float(lr))
    fitter = _FITTERS.get(key)
    if fitter is None:
        while len(_FITTERS) >= _fitters_max():
            _FITTERS.popitem(last=False)              # least recently used
        fitter = _FITTERS[key] = PoseFitter(vertices, faces, n, size, lr)
    else:
        _FITTERS.move_to_end(key)
    return fitter


def find_optimal_pose(vertices, faces, mask, bbox, square_bbox, image_size, K=None, num_iterations=50,
                      num_initializations=2000, lr=1e-2, image=None, debug=False, viz_folder="tmp", viz_step=10,
                      sort_best=True, rotations_init=None, viz=False, rend_size=constants.REND_SIZE, mode="auto"):
    """reference homan/pose_optimization.py:219-383 (debug plots not provided: `debug` / `viz` / `image` are accepted
    and ignored).  Returns the PoseOptimizer whose `rotations` / `translations` hold the best-ever pose first, then the
    poses sorted by final loss.
    mode="auto" (default) = "fused": the step as a fixed C-ABI launch sequence without the autograd tape, in a hipGraph
    (`_FusedPoseLoop`), run by a resident `PoseFitter` kept per (mesh, candidates, mask size, lr) - HOMAN_POSE_FITTER=0
    builds everything anew per call;
    mode="eager": the reference loop verbatim (torch autograd + Adam, one host sync per step for the best-ever bookkeeping);
    mode="graph": that same autograd step captured once in a hipGraph and replayed."""
    dev = torch.device("cuda")
    mesh_key = ((_digest(vertices), _digest(faces)) if torch.is_tensor(vertices) and torch.is_tensor(faces) else None)     # (the caller's tensors, before the device copies below)
    vertices = torch.as_tensor(vertices).float().to(dev)
    faces = torch.as_tensor(faces).to(dev)
    x, y, b, _ = [float(t) for t in square_bbox]
    K = torch.as_tensor(K, dtype=torch.float32)
    camintr_roi = get_K_crop_resize(K.unsqueeze(0), torch.tensor([[x, y, x + b, y + b]]), [rend_size]).to(dev)
    Kb = K.unsqueeze(0).to(dev)
    if rotations_init is None:
        rotations_init = compute_random_rotations(num_initializations, dev)
    rotations_init = torch.as_tensor(rotations_init).float().to(dev)
    translations_init = TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(vertices.unsqueeze(0), rotations_init),
                                                          Kb).unsqueeze(1)
    camintr_roi[:, :2] = camintr_roi[:, :2] / rend_size          # crop K to normalised rendering space (:321)
    if mode == "auto":
        mode = "fused"
    if mode not in ("eager", "graph", "fused"):
        raise ValueError(f"mode {mode} not in [auto|fused|eager|graph]")
    fitter = None
    if mode == "fused" and num_iterations > 0 and np.asarray(mask).shape[0] % 2 == 0:
        fitter = _resident_fitter(vertices, faces, num_initializations, int(np.asarray(mask).shape[0]), lr, mesh_key)
    if fitter is not None:
        model, final_losses, champion_rot, champion_trans = fitter.fit(mask, matrix_to_rot6d(rotations_init), translations_init,
                                                                       camintr_roi, num_iterations)
        _install_ranked_poses(model, final_losses, champion_rot, champion_trans, sort_best)
        return model
    model = PoseOptimizer(ref_image=mask, vertices=vertices, faces=faces, rotation_init=matrix_to_rot6d(rotations_init),
                          translation_init=translations_init, num_initializations=num_initializations, K=camintr_roi)
    if mode == "fused" and num_iterations > 0:
        final_losses, champion_rot, champion_trans = _fused_loop(model, lr, num_iterations)
    elif mode == "graph" and num_iterations > 0:
        final_losses, champion_rot, champion_trans = _graph_loop(model, lr, num_iterations)
    else:
        final_losses, champion_rot, champion_trans = _host_loop(model, lr, num_iterations)
    _install_ranked_poses(model, final_losses, champion_rot, champion_trans, sort_best)
    return model


def _host_loop(model, lr, num_iterations):
    """The reference's fit loop (:330-357) with torch Adam, one host round trip per step for the champion test.  The
    champion is the pose with the smallest loss EVER evaluated (strict `<`), recorded as it stands after the optimiser step
    that followed that evaluation (:348-353 copy the parameters behind `optimizer.step()`)."""
    adam = torch.optim.Adam(model.parameters(), lr=lr)
    champion = dict(loss=float("inf"), rot=None, trans=None)
    per_pose = None
    for _ in range(num_iterations):
        adam.zero_grad()
        terms, _, _ = model()
        per_pose = sum(terms.values())
        per_pose.sum().backward()
        adam.step()
        value, where = per_pose.detach().min(0)
        if float(value) < champion["loss"]:
            champion = dict(loss=float(value), rot=model.rotations.detach()[where].clone(),
                            trans=model.translations.detach()[where].clone())
    return per_pose, champion["rot"], champion["trans"]


def _install_ranked_poses(model, final_losses, champion_rot, champion_trans, ranked):
    """What the caller reads back (:359-381): unranked = the candidates in their original order (`find_optimal_poses`
    needs them aligned from frame to frame); ranked = the champion in slot 0, then the candidates by ascending final loss
    with the last one dropped so that the count stays `num_initializations`."""
    rots, trans = model.rotations.detach(), model.translations.detach()
    if ranked:
        order = torch.argsort(final_losses.detach())[:-1]
        rots = torch.cat([champion_rot[None], rots[order]])
        trans = torch.cat([champion_trans[None], trans[order]])
    model.rotations = nn.Parameter(rots.clone())
    model.translations = nn.Parameter(trans.clone())


def rot6d_to_matrix(rot_6d):
    """reference homan/utils/geometry.py:9-27 (3x2 -> 3x3, Gram-Schmidt, cross product per rotation); plain torch: this is
    result formatting of a handful of matrices, not path arithmetic (the kernels convert in csrc/hm_common.h)."""
    rot_6d = rot_6d.view(-1, 3, 2)
    a1, a2 = rot_6d[:, :, 0], rot_6d[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum("bi,bi->b", b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def find_optimal_poses(image_size, faces=None, vertices=None, annotations=None, images=None, Ks=None, num_iterations=50,
                       num_initializations=2000, viz_path="tmp.png", debug=False, rend_size=constants.REND_SIZE,
                       mode="auto"):
    """reference homan/pose_optimization.py:386-488 - the entry point of fit_vid_dataset.py:285-296.  One
    `find_optimal_pose` fit per frame, every frame started from the previous frame's `num_initializations` rotations
    (`sort_best=False` keeps the candidates aligned across frames); the motion kept is the candidate with the highest mean
    IoU over the clip (:468).  annotations[i]: {"target_crop_mask" (S,S) ndarray in {-1,0,1}, "bbox" xywh, "square_bbox"
    xywh, "full_mask" tensor}.  Returns one dict per frame: rotations (1,3,3), translations (1,1,3), verts_trans (1,V,3),
    target_masks (1,S,S), K_roi (1,1,3,3), masks, verts (1,V,3), full_mask."""
    mesh_v, mesh_f = torch.as_tensor(vertices), torch.as_tensor(faces)
    if mesh_v.dim() != 2 or mesh_v.shape[1] != 3 or mesh_f.dim() != 2 or mesh_f.shape[1] != 3:
        raise AssertionError("vertices (V,3) and faces (F,3) of ONE mesh expected")
    dev = torch.device("cuda")
    mesh_v, mesh_f = mesh_v.float().to(dev), mesh_f.to(dev)
    frames = len(annotations)
    pictures = list(images) if images is not None else [None] * frames
    # pass 1: frame by frame, all candidates kept side by side (frame t starts from frame t-1's fitted rotations)
    cand_rot, cand_trans, cand_verts, cand_iou, roi_K = [], [], [], [], []
    seed_rotations = None
    for t in range(frames):
        ann = annotations[t]
        fit = find_optimal_pose(vertices=mesh_v, faces=mesh_f, image=pictures[t], mask=ann["target_crop_mask"],
                                bbox=ann["bbox"], square_bbox=ann["square_bbox"], image_size=image_size, K=Ks[t],
                                num_iterations=num_iterations, num_initializations=num_initializations, debug=debug,
                                sort_best=False, rotations_init=seed_rotations, rend_size=rend_size, mode=mode)
        with torch.no_grad():
            cand_iou.append(fit()[1].detach())
            cand_verts.append(fit.apply_transformation().detach())
            seed_rotations = rot6d_to_matrix(fit.rotations.detach())
        cand_rot.append(seed_rotations)
        cand_trans.append(fit.translations.detach())
        roi_K.append(fit.K.detach())
    # pass 2: ONE candidate index for the whole clip - the one whose IoU, averaged over the frames, is highest (:468)
    # (last entry of the ascending argsort, as the reference takes it: same pick as it on ties, e.g. all-zero IoUs)
    winner = int(torch.argsort(torch.stack(cand_iou).mean(0))[-1])
    results = []
    for t in range(frames):
        ann = annotations[t]
        full = torch.as_tensor(ann["full_mask"]).to(dev)
        results.append({"rotations": cand_rot[t][winner][None], "translations": cand_trans[t][winner][None],
                        "verts_trans": cand_verts[t][winner][None],
                        "target_masks": torch.from_numpy(np.asarray(ann["target_crop_mask"])).to(dev)[None],
                        "K_roi": roi_K[t][None], "masks": full[None], "verts": mesh_v[None], "full_mask": full})
    return results
