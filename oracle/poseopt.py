"""CPU restatement of the object-pose initialisation of the reference.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows reference homan/pose_optimization.py:37-160 (PoseOptimizer) and
:219-383 (find_optimal_pose, without its debug plots), homan/lib3d/optitrans.py:29-127 and
homan/utils/geometry.py:89-134; the renderer leaf is oracle/nmr.py (anti_aliasing=False), the crop-intrinsics leaf is
oracle/yana.get_K_crop_resize (third-party libyana, PARITY UNPINNED).  Pinned to the reference's own code by the
`ref_poseinit_*` goldens (tools/refharness/gen_goldens.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn
from scipy.ndimage import distance_transform_edt

from . import nmr, yana
from .model import _rowvec_times_matrix, matrix_to_rot6d, rot6d_to_matrix

REND_SIZE = 256          # reference homan/constants.py


def compute_random_rotations(B=10):
    """utils/geometry.py:89-134, upright=False branch (Arvo, "Fast Random Rotation Matrices", 1992)."""
    x1, x2, x3 = torch.split(torch.rand(3 * B), B)
    tau = 2 * math.pi
    zeros, ones = torch.zeros_like(x1), torch.ones_like(x1)
    R = torch.stack((torch.stack((torch.cos(tau * x1), torch.sin(tau * x1), zeros), 1),
                     torch.stack((-torch.sin(tau * x1), torch.cos(tau * x1), zeros), 1),
                     torch.stack((zeros, zeros, ones), 1)), 1)
    v = torch.stack((torch.cos(tau * x2) * torch.sqrt(x3), torch.sin(tau * x2) * torch.sqrt(x3), torch.sqrt(1 - x3)), 1)
    H = torch.eye(3).repeat(B, 1, 1) - 2 * v.unsqueeze(2) * v.unsqueeze(1)
    return -torch.matmul(H, R)


def TCO_init_from_boxes_zup_autodepth(boxes_2d, model_points_3d, K):
    """lib3d/optitrans.py:83-127: translation that matches the projected box of the points with an xywh box."""
    model_points_3d = yana.tensorify(model_points_3d)
    bsz = model_points_3d.shape[0]
    K = yana.tensorify(K).float()
    boxes_2d = yana.tensorify(boxes_2d).float()
    if boxes_2d.dim() == 1:
        boxes_2d = boxes_2d.unsqueeze(0)
    if boxes_2d.shape[0] != bsz:
        boxes_2d = boxes_2d.repeat(bsz, 1)
    if K.dim() == 2:
        K = K.unsqueeze(0)
    if K.shape[0] != bsz:
        K = K.repeat(bsz, 1, 1)
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
        C_pts_3d = model_points_3d + trans.unsqueeze(1)
        proj_pts = yana.batch_proj2d(C_pts_3d, K)
        diag_proj = (proj_pts.min(1)[0] - proj_pts.max(1)[0]).norm(2, -1)
        proj_xy_centers = (proj_pts.min(1)[0] + proj_pts.max(1)[0]) / 2
        delta_z = z * (diag_proj / diag_bb - 1).unsqueeze(-1)
        z = z + delta_z
        xy_init = xy_init + ((bb_xy_centers - proj_xy_centers) * z) / fxfy
        trans = torch.cat([xy_init, z], 1)
    return trans


class PoseOptimizer(nn.Module):
    """pose_optimization.py:37-160.  `render_fn(verts (N,V,3), faces (N,F,3), K (1,3,3), size) -> (N,size,size)` is the
    only leaf: hard silhouettes without anti-aliasing."""

    def __init__(self, ref_image, vertices, faces, rotation_init, translation_init, num_initializations=1, kernel_size=7,
                 K=None, power=0.25, lw_chamfer=0, render_fn=None, written_out=False):
        assert ref_image.shape[0] == ref_image.shape[1], "Must be square."
        super().__init__()
        # written_out: the rigid transform as ((x*R0j + y*R1j) + z*R2j) + t, one IEEE operation per product / sum (the order
        # the HIP kernels follow: coverage then agrees with them bit for bit).  Default: the reference's torch.matmul
        # (:109), whose rounding depends on the host BLAS - the form the reference-generated golden is compared in.
        self.written_out = bool(written_out)
        self.register_buffer("vertices", vertices.repeat(num_initializations, 1, 1))
        self.register_buffer("faces", faces.repeat(num_initializations, 1, 1))
        # Convention for the silhouette-aware loss: -1 = occlusion, 0 = background, 1 = foreground (:66-74)
        image_ref = torch.from_numpy((ref_image > 0).astype(np.float32))
        keep_mask = torch.from_numpy((ref_image >= 0).astype(np.float32))
        self.register_buffer("image_ref", image_ref.repeat(num_initializations, 1, 1))
        self.register_buffer("keep_mask", keep_mask.repeat(num_initializations, 1, 1))
        self.pool = torch.nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=(kernel_size // 2))
        self.rotations = nn.Parameter(rotation_init.clone().float(), requires_grad=True)
        if rotation_init.shape[0] != translation_init.shape[0]:
            translation_init = translation_init.repeat(num_initializations, 1, 1)
        self.translations = nn.Parameter(translation_init.clone().float(), requires_grad=True)
        mask_edge = self.compute_edges(image_ref.unsqueeze(0)).cpu().numpy()
        edt = distance_transform_edt(1 - (mask_edge > 0)) ** (power * 2)
        self.register_buffer("edt_ref_edge", torch.from_numpy(edt).repeat(num_initializations, 1, 1).float())
        if K is None:
            K = torch.tensor([[[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]]], dtype=torch.float32)
        self.image_size = ref_image.shape[0]
        self.lw_chamfer = lw_chamfer
        self.K = K
        self.far = nmr.DEFAULT_FAR
        self.render_fn = render_fn if render_fn is not None else oracle_render

    def apply_transformation(self):
        if self.written_out:
            return _rowvec_times_matrix(self.vertices, rot6d_to_matrix(self.rotations)) + self.translations
        return torch.matmul(self.vertices, rot6d_to_matrix(self.rotations)) + self.translations

    def compute_offscreen_loss(self, verts):
        """:112-135: on-screen = NDC xy in [-1,1] and 0 < depth < far."""
        proj = nmr.projection(verts, self.K.to(verts.device), torch.eye(3, device=verts.device)[None],
                              torch.zeros(1, 3, device=verts.device), torch.zeros(1, 5, device=verts.device), 1)
        coord_xy, coord_z = proj[:, :, :2], proj[:, :, 2:]
        zeros = torch.zeros_like(coord_z)
        lower_right = torch.max(coord_xy - 1, zeros).sum(dim=(1, 2))
        upper_left = torch.max(-1 - coord_xy, zeros).sum(dim=(1, 2))
        behind = torch.max(-coord_z, zeros).sum(dim=(1, 2))
        too_far = torch.max(coord_z - self.far, zeros).sum(dim=(1, 2))
        return lower_right + upper_left + behind + too_far

    def compute_edges(self, silhouette):
        return self.pool(silhouette) - silhouette

    def forward(self):
        verts = self.apply_transformation()
        image = self.keep_mask * self.render_fn(verts, self.faces, self.K, self.image_size)
        loss_dict = {}
        loss_dict["mask"] = torch.sum((image - self.image_ref) ** 2, dim=(1, 2))
        with torch.no_grad():
            iou = yana.batch_mask_iou(image.detach(), self.image_ref.detach())
        loss_dict["chamfer"] = self.lw_chamfer * torch.sum(self.compute_edges(image) * self.edt_ref_edge, dim=(1, 2))
        loss_dict["offscreen"] = 100000 * self.compute_offscreen_loss(verts)
        return loss_dict, iou, image


def oracle_render(verts, faces, K, size):
    r = nmr.Renderer(image_size=size, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1, anti_aliasing=False)
    return r(verts, faces, mode="silhouettes")


def find_optimal_pose(vertices, faces, mask, bbox, square_bbox, image_size, K=None, num_iterations=50,
                      num_initializations=2000, lr=1e-2, sort_best=True, rotations_init=None, render_fn=None,
                      device="cpu", rend_size=REND_SIZE):
    """pose_optimization.py:219-383 without the debug plots: sample / take rotations, closed-form translation init,
    `num_iterations` Adam steps on all initialisations at once, best-ever pose first, then sorted by final loss."""
    x, y, b, _ = square_bbox
    camintr_roi = yana.get_K_crop_resize(torch.as_tensor(K, dtype=torch.float32).unsqueeze(0),
                                         torch.tensor([[x, y, x + b, y + b]], dtype=torch.float32), [rend_size])
    K = yana.tensorify(K).float().unsqueeze(0)
    if rotations_init is None:
        rotations_init = compute_random_rotations(num_initializations)
    translations_init = TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(vertices.unsqueeze(0), rotations_init),
                                                          K).unsqueeze(1)
    camintr_roi[:, :2] = camintr_roi[:, :2] / rend_size          # crop K to normalised rendering space (:321)
    model = PoseOptimizer(ref_image=mask, vertices=vertices, faces=faces, rotation_init=matrix_to_rot6d(rotations_init),
                          translation_init=translations_init, num_initializations=num_initializations, K=camintr_roi,
                          render_fn=render_fn).to(device)
    model.K = model.K.to(device)
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


def find_optimal_poses(image_size, faces=None, vertices=None, annotations=None, images=None, Ks=None, num_iterations=50,
                       num_initializations=2000, rend_size=REND_SIZE):
    """pose_optimization.py:386-488: per-frame fits chained through the previous frame's rotations (sort_best=False), the
    candidate with the highest mean IoU over the clip kept (:468)."""
    vertices, faces = torch.as_tensor(vertices).float(), torch.as_tensor(faces)
    previous_rotations, all_object_parameters, all_losses = None, [], []
    for annotation, K in zip(annotations, Ks):
        model = find_optimal_pose(vertices, faces, annotation["target_crop_mask"], annotation["bbox"],
                                  annotation["square_bbox"], image_size, K=K, num_iterations=num_iterations,
                                  num_initializations=num_initializations, sort_best=False,
                                  rotations_init=previous_rotations, rend_size=rend_size)
        with torch.no_grad():
            _, iou, _ = model()
            verts_trans = model.apply_transformation()
        rotations = rot6d_to_matrix(model.rotations.detach())
        all_object_parameters.append({
            "rotations": rotations, "translations": model.translations.detach(),
            "target_masks": torch.from_numpy(np.asarray(annotation["target_crop_mask"])), "K_roi": model.K.detach(),
            "masks": torch.as_tensor(annotation["full_mask"]), "verts": vertices.detach(),
            "verts_trans": verts_trans.detach()})
        previous_rotations = rotations
        all_losses.append(iou.detach())
    all_losses = torch.stack(all_losses)
    best_idx = torch.argsort(all_losses.mean(0))[-1]
    out = []
    for obj_params, info in zip(all_object_parameters, annotations):
        final_params = {key: obj_params[key][best_idx].unsqueeze(0) for key in ("rotations", "translations", "verts_trans")}
        for key in ("target_masks", "K_roi", "masks", "verts"):
            final_params[key] = obj_params[key].unsqueeze(0)
        final_params["full_mask"] = torch.as_tensor(info["full_mask"])
        out.append(final_params)
    return out
