"""Object-pose initialisation from an instance mask on the HIP rasteriser.

Mirrors reference homan/pose_optimization.py: `PoseOptimizer` (:37-160, same constructor keywords minus `textures`,
same parameter names `rotations` / `translations`, same `forward() -> (loss_dict, iou, image)`) and
`find_optimal_pose` (:219-383, same arguments and returned module; the debug plots are not provided).  The render leaf
nr.Renderer(image_size, anti_aliasing=False) is `ops.silhouette_render_noaa` (csrc/raster.hip, hm_sil_fwd `alpha_full`
+ hm_sil_bwd mode 3); everything else is host logic in torch, as in the reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from scipy.ndimage import distance_transform_edt

from . import constants, ops
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


def TCO_init_from_boxes_zup_autodepth(boxes_2d, model_points_3d, K):
    """reference homan/lib3d/optitrans.py:83-127: translation matching the projected box of the points with an xywh box."""
    model_points_3d = torch.as_tensor(model_points_3d)
    bsz, device = model_points_3d.shape[0], model_points_3d.device
    K = torch.as_tensor(K, dtype=torch.float32).to(device)
    boxes_2d = torch.as_tensor(boxes_2d, dtype=torch.float32).to(device)
    if boxes_2d.dim() == 1:
        boxes_2d = boxes_2d.unsqueeze(0)
    if boxes_2d.shape[0] != bsz:
        boxes_2d = boxes_2d.repeat(bsz, 1)
    if K.dim() == 2:
        K = K.unsqueeze(0)
    if K.shape[0] != bsz:
        K = K.repeat(bsz, 1, 1)
    assert boxes_2d.shape[-1] == 4 and boxes_2d.dim() == 2
    boxes_2d = torch.stack([boxes_2d[:, 0], boxes_2d[:, 1], boxes_2d[:, 0] + boxes_2d[:, 2],
                            boxes_2d[:, 1] + boxes_2d[:, 3]], 1)
    diag_bb = (boxes_2d[:, [2, 3]] - boxes_2d[:, [0, 1]]).norm(2, -1)
    bb_xy_centers = (boxes_2d[:, [0, 1]] + boxes_2d[:, [2, 3]]) / 2
    fxfy = K[:, [0, 1], [0, 1]]
    cxcy = K[:, [0, 1], [2, 2]]
    z = fxfy.new_ones(bsz, 1)
    xy_init = ((bb_xy_centers - cxcy) * z) / fxfy
    trans = torch.cat([xy_init, z], 1)
    for _ in range(10):
        pts = model_points_3d + trans.unsqueeze(1)
        hom = K.bmm(pts.transpose(1, 2)).transpose(1, 2)
        proj_pts = hom[:, :, :2] / hom[:, :, 2:]
        diag_proj = (proj_pts.min(1)[0] - proj_pts.max(1)[0]).norm(2, -1)
        proj_xy_centers = (proj_pts.min(1)[0] + proj_pts.max(1)[0]) / 2
        z = z + z * (diag_proj / diag_bb - 1).unsqueeze(-1)
        xy_init = xy_init + ((bb_xy_centers - proj_xy_centers) * z) / fxfy
        trans = torch.cat([xy_init, z], 1)
    return trans


class PoseOptimizer(nn.Module):
    """reference homan/pose_optimization.py:37-160 (occlusion-aware silhouette loss + one-way edge chamfer + off-screen
    penalty over `num_initializations` candidate poses of one mesh against one instance mask)."""

    def __init__(self, ref_image, vertices, faces, rotation_init, translation_init, num_initializations=1, kernel_size=7,
                 K=None, power=0.25, lw_chamfer=0, textures=None):
        assert ref_image.shape[0] == ref_image.shape[1], "Must be square."
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("homan_amd.pose_optimization needs an MI355X (ROCm) device; there is no CPU path")
        dev = torch.device("cuda")
        size = int(ref_image.shape[0])
        if size % 32:
            raise NotImplementedError(f"the rasteriser needs image sizes that are multiples of 32, got {size}")
        # (every (N,...) buffer is replicated ON the device: the reference builds the 500-fold copies of the mask on the host
        #  and uploads ~400 MB per fit, which was two thirds of a whole 50-step fit here)
        vertices, faces = torch.as_tensor(vertices).float().to(dev), torch.as_tensor(faces).to(dev)
        self.register_buffer("vertices", vertices.repeat(num_initializations, 1, 1))
        self.register_buffer("faces", faces.repeat(num_initializations, 1, 1))
        # Convention for the silhouette-aware loss: -1 = occlusion, 0 = background, 1 = foreground (:66-74)
        ref_image = np.asarray(ref_image)
        image_ref = torch.from_numpy((ref_image > 0).astype(np.float32)).to(dev)
        keep_mask = torch.from_numpy((ref_image >= 0).astype(np.float32)).to(dev)
        self.register_buffer("image_ref", image_ref.repeat(num_initializations, 1, 1))
        self.register_buffer("keep_mask", keep_mask.repeat(num_initializations, 1, 1))
        self.pool = torch.nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=(kernel_size // 2))
        self.rotations = nn.Parameter(torch.as_tensor(rotation_init).clone().float().to(dev), requires_grad=True)
        translation_init = torch.as_tensor(translation_init).to(dev)
        if rotation_init.shape[0] != translation_init.shape[0]:
            translation_init = translation_init.repeat(num_initializations, 1, 1)
        self.translations = nn.Parameter(translation_init.clone().float(), requires_grad=True)
        mask_edge = self.compute_edges(image_ref.unsqueeze(0)).cpu().numpy()
        edt = distance_transform_edt(1 - (mask_edge > 0)) ** (power * 2)
        self.register_buffer("edt_ref_edge", torch.from_numpy(edt).float().to(dev).repeat(num_initializations, 1, 1))
        if K is None:
            K = torch.tensor([[[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]]], dtype=torch.float32)
        self.register_buffer("K", torch.as_tensor(K).float().reshape(-1, 3, 3)[:1].clone())
        self.image_size, self.lw_chamfer = size, lw_chamfer
        self.to(dev)
        n = self.vertices.shape[0]
        self._one = torch.ones(1, device=dev)
        self._keep1, self._ref1 = self.keep_mask[0].contiguous(), self.image_ref[0].contiguous()
        self._K_all = self.K.repeat(n, 1, 1).contiguous()
        self._sil_ctx = ops.SilhouetteContext(self.faces, self.vertices.shape[1], n, size // 2, dev)

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
        coord_xy = torch.stack([2 * (u - 0.5), 2 * (v - 0.5)], -1)
        coord_z = z.unsqueeze(-1)
        zeros = torch.zeros_like(coord_z)
        lower_right = torch.max(coord_xy - 1, zeros).sum(dim=(1, 2))
        upper_left = torch.max(-1 - coord_xy, zeros).sum(dim=(1, 2))
        behind = torch.max(-coord_z, zeros).sum(dim=(1, 2))
        too_far = torch.max(coord_z - NMR_FAR, zeros).sum(dim=(1, 2))
        return lower_right + upper_left + behind + too_far

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
        graph = torch.cuda.CUDAGraph()
        for p in params:
            p.grad.zero_()
        with torch.cuda.graph(graph):
            step()
            for p in params:
                p.grad.zero_()
        for _ in range(num_iterations - done):       # (capturing records the step, it does not run it)
            graph.replay()
    torch.cuda.synchronize()
    return losses_out, best_rot.clone(), best_trans.clone()


def find_optimal_pose(vertices, faces, mask, bbox, square_bbox, image_size, K=None, num_iterations=50,
                      num_initializations=2000, lr=1e-2, image=None, debug=False, viz_folder="tmp", viz_step=10,
                      sort_best=True, rotations_init=None, viz=False, rend_size=constants.REND_SIZE, mode="eager"):
    """reference homan/pose_optimization.py:219-383 (debug plots not provided: `debug` / `viz` / `image` are accepted
    and ignored).  Returns the PoseOptimizer whose `rotations` / `translations` hold the best-ever pose first, then the
    poses sorted by final loss.
    mode="eager": the reference loop verbatim (one host sync per step for the best-ever bookkeeping);
    mode="graph": the same step (forward, backward, Adam, best-ever bookkeeping on the device) captured once in a hipGraph
    and replayed (measured: no faster than the eager loop here, the step is GPU-bound; kept for host-bound setups)."""
    dev = torch.device("cuda")
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
    model = PoseOptimizer(ref_image=mask, vertices=vertices, faces=faces, rotation_init=matrix_to_rot6d(rotations_init),
                          translation_init=translations_init, num_initializations=num_initializations, K=camintr_roi)
    if mode == "graph" and num_iterations > 0:
        losses, best_rots_single, best_trans_single = _graph_loop(model, lr, num_iterations)
    elif mode in ("eager", "graph"):
        optimizer = torch.optim.Adam(model.parameters(), lr=lr)
        best_loss_single, best_rots_single, best_trans_single = np.inf, None, None
        for _ in range(num_iterations):
            optimizer.zero_grad()
            loss_dict, _iou, _sil = model()
            losses = sum(loss_dict.values())
            losses.sum().backward()
            optimizer.step()
            if losses.min() < best_loss_single:
                ind = torch.argmin(losses)
                best_loss_single = losses[ind]
                best_rots_single = model.rotations[ind].detach().clone()
                best_trans_single = model.translations[ind].detach().clone()
    else:
        raise ValueError(f"mode {mode} not in [eager|graph]")
    best_rots, best_trans, best_losses = model.rotations, model.translations, losses
    if sort_best:
        inds = torch.argsort(best_losses)
        best_trans = best_trans[inds][:num_initializations].detach().clone()
        best_rots = best_rots[inds][:num_initializations].detach().clone()
        best_rots = torch.cat((best_rots_single.unsqueeze(0), best_rots[:-1]), 0)
        best_trans = torch.cat((best_trans_single.unsqueeze(0), best_trans[:-1]), 0)
    model.rotations = nn.Parameter(best_rots)
    model.translations = nn.Parameter(best_trans)
    return model


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
                       mode="eager"):
    """reference homan/pose_optimization.py:386-488 - the entry point of fit_vid_dataset.py:285-296.  One
    `find_optimal_pose` fit per frame, every frame started from the previous frame's `num_initializations` rotations
    (`sort_best=False` keeps the candidates aligned across frames); the motion kept is the candidate with the highest mean
    IoU over the clip (:468).  annotations[i]: {"target_crop_mask" (S,S) ndarray in {-1,0,1}, "bbox" xywh, "square_bbox"
    xywh, "full_mask" tensor}.  Returns one dict per frame: rotations (1,3,3), translations (1,1,3), verts_trans (1,V,3),
    target_masks (1,S,S), K_roi (1,1,3,3), masks, verts (1,V,3), full_mask."""
    vertices, faces = torch.as_tensor(vertices), torch.as_tensor(faces)
    assert vertices.dim() == 2 and vertices.shape[1] == 3 and faces.dim() == 2 and faces.shape[1] == 3
    dev = torch.device("cuda")
    vertices, faces = vertices.float().to(dev), faces.to(dev)
    previous_rotations, all_object_parameters, all_losses = None, [], []
    images = images if images is not None else [None] * len(annotations)
    for image, annotation, K in zip(images, annotations, Ks):
        model = find_optimal_pose(vertices=vertices, faces=faces, image=image, mask=annotation["target_crop_mask"],
                                  bbox=annotation["bbox"], square_bbox=annotation["square_bbox"], image_size=image_size,
                                  K=K, num_iterations=num_iterations, num_initializations=num_initializations, debug=debug,
                                  sort_best=False, rotations_init=previous_rotations, rend_size=rend_size, mode=mode)
        with torch.no_grad():
            _, iou, _ = model()
            verts_trans = model.apply_transformation()
            rotations = rot6d_to_matrix(model.rotations.detach())
        all_object_parameters.append({
            "rotations": rotations, "translations": model.translations.detach(),
            "target_masks": torch.from_numpy(np.asarray(annotation["target_crop_mask"])).to(dev),
            "K_roi": model.K.detach(), "masks": torch.as_tensor(annotation["full_mask"]).to(dev),
            "verts": vertices.detach(), "verts_trans": verts_trans.detach()})
        previous_rotations = rotations
        all_losses.append(iou.detach())
    all_losses = torch.stack(all_losses)                          # (frame_nb, num_initializations)
    best_idx = torch.argsort(all_losses.mean(0))[-1]              # highest mean IoU over the sequence
    all_final_params = []
    for obj_params, info in zip(all_object_parameters, annotations):
        final_params = {key: obj_params[key][best_idx].unsqueeze(0) for key in ("rotations", "translations", "verts_trans")}
        for key in ("target_masks", "K_roi", "masks", "verts"):
            final_params[key] = obj_params[key].unsqueeze(0)
        final_params["full_mask"] = torch.as_tensor(info["full_mask"]).to(dev)
        all_final_params.append(final_params)
    return all_final_params
