"""HOMan -- the hand-object model of the reference (reference homan/homan.py:26-508), MI355X-native.

Same constructor keywords, same Parameter / buffer names (so `named_parameters()` filtering by "mano" /
"rotation" in reference homan/jointopt.py:128-151 and `load_state_dict(strict=False)` of a reference
`joint_fit.pt` behave identically), same `forward(loss_weights) -> (loss_dict, metric_dict)`,
`get_verts_object()` and `get_verts_hand(detach_scale=False)`.  All arithmetic runs in the hand-written HIP kernels
of libhoman_amd.so through `homan_amd.ops`; there is no CPU path.

Visualisation (off the optimisation path, no gradients): `model.renderer` (homan_amd.nmr.Renderer with the reference's
light, homan.py:168-176), the combined scene topology / colour buffers (:177-219) and render / render_gt /
render_with_gt / render_limem / save_obj (:510-628) on the same rasteriser.  Not mirrored: assign_human_masks (:239-296,
never called by the reference).
"""
import numpy as np
import torch
from torch import nn

from . import constants, lossutils, nmr, ops, trans3d
from .losses import Losses
from .manomodel import ManoModel
from .meshutils import get_faces_and_textures


def combine_verts(verts_list):
    """reference homan/utils/geometry.py:43-47."""
    batch_size = verts_list[0].shape[0]
    return torch.cat([v.reshape(batch_size, -1, 3) for v in verts_list], 1)


def matrix_to_rot6d(rotmat):
    """reference homan/utils/geometry.py:30-40."""
    return rotmat.view(-1, 3, 3)[:, :, :2]


def weakcam_persp_trans(cams, K, image_size=640, reference_depth=1.0):
    """Camera-space translation (B,1,3) of the scaled-orthographic hand cameras `cams` = [s, tx, ty] (reference
    homan/utils/camera.py:85-97: the camera goes to pixel units of a 640-pixel image - `compute_transformation_ortho`'s default
    `image_size`, which HOMan.get_verts_hand does not override - and through `libyana.camutils.camconvs.
    batch_weakcam2persptrans(cams_px, K_px, 1)`).  libyana is not in /root/reference: the conversion is the first-order
    identity between the two camera models (u = s X + t  vs  u = f (X + T) / (Z + T_z) + c  =>  T_z = f / s,
    T_xy = (t - c) T_z / f_xy), restated from the model, not from its source (parity unpinned; oracle/yana.py states the
    same)."""
    persp_scale = cams[:, :1] / 2 * image_size
    persp_trans = (cams[:, 1:] + 1 / cams[:, :1]) * persp_scale
    fx, fy = K[:, 0, 0] * image_size, K[:, 1, 1] * image_size
    cx, cy = K[:, 0, 2] * image_size, K[:, 1, 2] * image_size
    tz = reference_depth * fx / persp_scale[:, 0]
    tx = (persp_trans[:, 0] - cx) * tz / fx
    ty = (persp_trans[:, 1] - cy) * tz / fy
    return torch.stack([tx, ty, tz.expand_as(tx)], 1).unsqueeze(1)


class HOMan(nn.Module):
    def __init__(self, translations_object, rotations_object, verts_object_og, faces_object, translations_hand,
                 rotations_hand, verts_hand_og, ref_verts2d_hand, hand_sides, mano_trans, mano_rot, mano_betas,
                 mano_pca_pose, faces_hand, masks_object, masks_hand, camintr_rois_object, camintr_rois_hand,
                 target_masks_object, target_masks_hand, class_name, cams_hand=None, int_scale_init=1.0, camintr=None,
                 optimize_object_scale=False, optimize_ortho_cam=True, hand_proj_mode="persp", optimize_mano=True,
                 optimize_mano_beta=True, inter_type="centroid", image_size=640,
                 # homan_amd extensions (keyword-only use)
                 mano_model=None, mano_root="extra_data/mano", rend_size=constants.REND_SIZE, sync_metrics=True,
                 ordinal_depth=False):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("homan_amd.HOMan needs an MI355X (ROCm) device; there is no CPU path")
        dev = torch.device("cuda")
        f32 = lambda t: t.detach().clone().float()

        self.translations_object = nn.Parameter(f32(translations_object), requires_grad=True)
        self.hand_proj_mode = hand_proj_mode
        rotations_object = f32(rotations_object)
        rot6d_o = matrix_to_rot6d(rotations_object) if rotations_object.shape[-1] == 3 else rotations_object
        self.rotations_object = nn.Parameter(rot6d_o.detach().clone().contiguous(), requires_grad=True)
        self.register_buffer("verts_object_og", f32(verts_object_og))

        self.translations_hand = nn.Parameter(f32(translations_hand), requires_grad=True)
        rotations_hand = f32(rotations_hand)
        if rotations_hand.shape[-1] == 3:
            rotations_hand = matrix_to_rot6d(rotations_hand)
        self.rotations_hand = nn.Parameter(rotations_hand.detach().clone().contiguous(), requires_grad=True)
        if cams_hand is None:
            if hand_proj_mode == "ortho":      # (checked here, on the host input: forward may run inside a stream capture)
                raise ValueError("hand_proj_mode='ortho' needs cams_hand (scaled-orthographic [s, tx, ty] per hand, s != 0)")
            cams_hand = torch.zeros(translations_hand.shape[0], 3)
        elif hand_proj_mode == "ortho" and bool((torch.as_tensor(cams_hand).detach().reshape(-1, 3)[:, 0] == 0).any()):
            raise ValueError("hand_proj_mode='ortho': cams_hand holds a zero scale (the translation is f / s)")
        if optimize_ortho_cam:
            self.cams_hand = nn.Parameter(f32(cams_hand), requires_grad=True)
        else:
            self.register_buffer("cams_hand", f32(cams_hand))
        self.hand_sides = hand_sides
        self.hand_nb = len(hand_sides)
        for side in hand_sides:
            if side not in ("right", "left"):
                raise ValueError(f"{side} not in [left|right]")           # reference manomodel.py:141
        if self.hand_nb not in (1, 2):      # the reference's collision term de-interleaves with a stride of 2 (lossutils.py:58)
            raise NotImplementedError("one or two hands per frame; got %d" % self.hand_nb)

        self.optimize_mano = optimize_mano
        if optimize_mano:
            self.mano_pca_pose = nn.Parameter(f32(mano_pca_pose), requires_grad=True)
            self.mano_rot = nn.Parameter(f32(mano_rot), requires_grad=True)
            self.mano_trans = nn.Parameter(f32(mano_trans), requires_grad=True)
        else:   # reference homan.py:104-106: no mano_trans at all in this mode
            self.register_buffer("mano_pca_pose", f32(mano_pca_pose))
            self.register_buffer("mano_rot", f32(mano_rot))
        if optimize_mano_beta:
            self.mano_betas = nn.Parameter(torch.zeros_like(f32(mano_betas)), requires_grad=True)
            self.register_buffer("int_scales_hand", torch.ones(1).float() * int_scale_init)
        else:
            self.register_buffer("mano_betas", torch.zeros_like(f32(mano_betas)))
            self.int_scales_hand = nn.Parameter(int_scale_init * torch.ones(1).float(), requires_grad=True)
        self.register_buffer("verts_hand_og", f32(verts_hand_og))
        self.register_buffer("ref_verts2d_hand", f32(ref_verts2d_hand))

        # (reference homan.py:122 wraps the argument in torch.Tensor(...), which raises for its own float default 1.0 -
        #  callers pass the int 1, jointopt.py:118; any real number is accepted here)
        init_scales = float(int_scale_init) * torch.ones(1).float()
        self.optimize_object_scale = optimize_object_scale
        if optimize_object_scale:
            self.int_scales_object = nn.Parameter(init_scales, requires_grad=True)
        else:
            self.register_buffer("int_scales_object", init_scales)
        self.register_buffer("int_scale_object_mean", torch.ones(1).float())
        self.register_buffer("int_scale_hand_mean", torch.ones(1).float())
        self.register_buffer("ref_mask_object", (target_masks_object > 0).float())
        self.register_buffer("keep_mask_object", (target_masks_object >= 0).float())
        self.register_buffer("ref_mask_hand", (target_masks_hand > 0).float())
        self.register_buffer("keep_mask_hand", (target_masks_hand >= 0).float())
        self.register_buffer("camintr_rois_object", f32(camintr_rois_object))
        self.register_buffer("camintr_rois_hand", f32(camintr_rois_hand))
        self.register_buffer("faces_object", faces_object.detach().clone())
        # white per-face textures of the single-mesh depth renders (reference homan.py:145-151; state_dict keys)
        self.register_buffer("textures_object", torch.ones(faces_object.shape[0], faces_object.shape[1], 1, 1, 1, 3))
        self.register_buffer("textures_hand", torch.ones(faces_hand.shape[0], faces_hand.shape[1], 1, 1, 1, 3))
        self.register_buffer("faces_hand", faces_hand.detach().clone())
        if camintr is None:
            camintr = torch.tensor([[[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]]], dtype=torch.float32)
        else:
            camintr = torch.as_tensor(camintr).float()
            if camintr.dim() == 2:
                camintr = camintr.unsqueeze(0)
        batch = translations_object.shape[0]
        if camintr.shape[0] == 1 and batch > 1:
            camintr = camintr.repeat(batch, 1, 1)
        self.register_buffer("camintr", camintr.contiguous())
        self.image_size = image_size
        if masks_hand is not None:
            self.register_buffer("masks_human", masks_hand.detach().clone())
        if masks_object.dim() == 2:
            masks_object = masks_object.unsqueeze(0)
        self.register_buffer("masks_object", masks_object.detach().clone())
        self.cuda()

        # leaves: MANO layer (kept out of saved checkpoints by its "mano_model" prefix, fit_vid_dataset.py:366-371)
        self.mano_model = ManoModel(mano_root, pca_comps=16, mano_model=mano_model, device=dev)
        self.sync_metrics = sync_metrics
        self.reduce_ws = ops.ReduceWorkspace(dev)
        num_verts_object = self.verts_object_og.shape[1]
        self.losses = Losses(renderer=None, ref_mask_object=self.ref_mask_object,
                             keep_mask_object=self.keep_mask_object, ref_mask_hand=self.ref_mask_hand,
                             ref_verts2d_hand=self.ref_verts2d_hand, keep_mask_hand=self.keep_mask_hand,
                             camintr_rois_object=self.camintr_rois_object, camintr_rois_hand=self.camintr_rois_hand,
                             camintr=self.camintr, class_name=class_name, hand_nb=self.hand_nb, inter_type=inter_type,
                             faces_object=self.faces_object, num_verts_object=num_verts_object, rend_size=rend_size,
                             reduce_ws=self.reduce_ws, sync_metrics=sync_metrics)
        closed = np.asarray(self.mano_model.closed_faces)
        if self.hand_nb == 1:
            self.collision_ctx = ops.CollisionContext(closed, self.faces_object[0], batch, 778, num_verts_object, dev)
        else:
            # scene [hand 0, hand 1, object], both hands with the closed topology in reversed winding (lossutils.py:53-59).
            # Every mesh's SDF depends on its own vertices only and the loss is a sum over ordered pairs of meshes
            # (scenesdf.py:131-146): three two-mesh launches give the same sum and the same gradients.
            rev = np.ascontiguousarray(closed[:, ::-1])
            rev_t = torch.as_tensor(rev.astype(np.int32))
            self.collision_ctx = (ops.CollisionContext(rev, rev_t, batch, 778, 778, dev),
                                  ops.CollisionContext(rev, self.faces_object[0], batch, 778, num_verts_object, dev),
                                  ops.CollisionContext(rev, self.faces_object[0], batch, 778, num_verts_object, dev))
        self._mano_cache = None
        # ordinal depth term: the reference's own call site cannot run (see forward()); `ordinal_depth=True` opts into the
        # loss the method describes (homan.py:384-419 + lossutils.py:133-169) instead of reproducing that TypeError
        self.ordinal_depth = bool(ordinal_depth)
        if self.ordinal_depth and int(image_size) > 1024:
            # hm_ordinal_depth_fwd packs a frame's three pixel counts into 21-bit fields of one 64-bit atomic: 1024^2 = 2^20
            # pixels per frame is its limit (the term is rendered at image_size, reference homan.py:168-172)
            raise NotImplementedError(f"ordinal depth term: image_size {image_size} > 1024 (rendered at image_size; scale the "
                                      "instance masks and intrinsics down, or leave ordinal_depth off)")
        self._depth_state = None
        self._depth_state_hands = None
        with torch.no_grad():
            self.verts_hand_init = self.get_verts_hand()[0].detach().clone()
            self.verts_object_init = self.get_verts_object()[0].detach().clone()
        self._setup_visualisation(batch)

    # ------------------------------------------------------------------ another clip into the same model
    def load_clip(self, translations_object, rotations_object, verts_object_og, translations_hand, rotations_hand,
                  verts_hand_og, ref_verts2d_hand, mano_trans, mano_rot, mano_betas, mano_pca_pose, masks_object, masks_hand,
                  camintr_rois_object, camintr_rois_hand, target_masks_object, target_masks_hand, cams_hand=None,
                  camintr=None, int_scale_init=1.0, faces_object=None, faces_hand=None, hand_sides=None, **_same):
        """The data arguments of the constructor (reference homan/homan.py:27-60) for ANOTHER clip of the same shapes, copied
        IN PLACE into this model's Parameters and buffers: every context, workspace and captured graph built on them stays
        valid (a resident stepper fits a stream of clips without rebuilding anything, jointopt.ClipFitter).  Same
        preprocessing as __init__: rotation matrices -> 6-D, betas start at zero (:108), scales at `int_scale_init`,
        target masks -> ref / keep (:131-134).  Topologies and hand sides are the model's own (checked when given)."""
        f32 = lambda t: torch.as_tensor(t).detach().float()

        def put(dst, src):
            src = f32(src)
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"load_clip: shape {tuple(src.shape)} does not fit the model's {tuple(dst.shape)}")
            dst.data.copy_(src, non_blocking=True)

        def rot6d(r):
            r = f32(r)
            return matrix_to_rot6d(r) if r.shape[-1] == 3 else r

        if hand_sides is not None and list(hand_sides) != list(self.hand_sides):
            raise ValueError("load_clip: other hand sides")
        if faces_object is not None and not torch.equal(torch.as_tensor(faces_object)[0].cpu(), self.faces_object[0].cpu()):
            raise ValueError("load_clip: other object topology")
        with torch.no_grad():
            put(self.translations_object, translations_object)
            put(self.rotations_object, rot6d(rotations_object))
            put(self.verts_object_og, verts_object_og)
            put(self.translations_hand, translations_hand)
            put(self.rotations_hand, rot6d(rotations_hand))
            put(self.cams_hand, cams_hand if cams_hand is not None else torch.zeros_like(self.cams_hand))
            put(self.mano_pca_pose, mano_pca_pose)
            put(self.mano_rot, mano_rot)
            if self.optimize_mano:
                put(self.mano_trans, mano_trans)
            self.mano_betas.data.zero_()
            self.int_scales_hand.data.fill_(float(int_scale_init))
            self.int_scales_object.data.fill_(float(int_scale_init))
            put(self.verts_hand_og, verts_hand_og)
            put(self.ref_verts2d_hand, ref_verts2d_hand)
            tmo = f32(target_masks_object).to(self.ref_mask_object.device, non_blocking=True)
            tmh = f32(target_masks_hand).to(self.ref_mask_hand.device, non_blocking=True)
            put(self.ref_mask_object, tmo > 0)
            put(self.keep_mask_object, tmo >= 0)
            put(self.ref_mask_hand, tmh > 0)
            put(self.keep_mask_hand, tmh >= 0)
            put(self.camintr_rois_object, camintr_rois_object)
            put(self.camintr_rois_hand, camintr_rois_hand)
            # (no intrinsics = the constructor's default, homan.py:113-116 - NOT the previous clip's: a clip loaded into a
            #  resident model must fit exactly like a freshly built one)
            c = (torch.tensor([[[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]]], dtype=torch.float32) if camintr is None
                 else torch.as_tensor(camintr).float())
            c = c.unsqueeze(0) if c.dim() == 2 else c
            put(self.camintr, c.expand_as(self.camintr) if c.shape[0] == 1 else c)
            self.losses.camintr.copy_(self.camintr)
            if masks_hand is not None and hasattr(self, "masks_human"):
                self.masks_human.copy_(torch.as_tensor(masks_hand).to(self.masks_human.dtype))
            mo = torch.as_tensor(masks_object)
            self.masks_object.copy_((mo.unsqueeze(0) if mo.dim() == 2 else mo).to(self.masks_object.dtype))
            self.losses.keep_sum.copy_(self.keep_mask_object.sum().reshape(1))
            self.losses.sil_ctx.invalidate_outputs()
            self._mano_cache = None
            if self._depth_state is not None:
                ctx_o, ctx_h, m_o, m_h = self._depth_state
                m_o.copy_((self.masks_object != 0).to(torch.uint8))
                m_h.copy_((self.masks_human != 0).to(torch.uint8))
            if getattr(self, "_depth_state_hands", None) is not None:
                _, _, m_o, m_hs = self._depth_state_hands
                m_o.copy_((self.masks_object != 0).to(torch.uint8))
                for h, m in enumerate(m_hs):
                    m.copy_((self.masks_human[h::self.hand_nb] != 0).to(torch.uint8))
            self.verts_hand_init.copy_(self.get_verts_hand()[0].detach())
            self.verts_object_init.copy_(self.get_verts_object()[0].detach())
            self._mano_cache = None

    # ------------------------------------------------------------------ visualisation (reference homan.py:168-219, 510-628)
    def _setup_visualisation(self, batch_size):
        """Renderer with the reference's light and the combined [object, hand...] scene meshes in prediction colours
        (gold / grey), ground-truth colours (green / blue) and both (:177-219).  Buffers are plain attributes, as in
        the reference (they are not part of the state_dict)."""
        self.renderer = nmr.Renderer(image_size=self.image_size, K=self.camintr.clone(), orig_size=1)
        self.renderer.light_direction = [1, 0.5, 1]
        self.renderer.light_intensity_direction = 0.3
        self.renderer.light_intensity_ambient = 0.5
        self.renderer.background_color = [1.0, 1.0, 1.0]
        # one scene = object first, then the hands (first frame's meshes give the per-mesh face counts).  Three colourings
        # of it, each stored as `faces<suffix>` / `textures<suffix>` replicated over the frames (:177-219): the fit
        # (gold object, grey hands), the ground truth (green, blue), and both in one scene (fit meshes, then truth meshes).
        scene = [(self.verts_object_init[:1], self.faces_object[:1])]
        scene += [(self.verts_hand_init[h:h + 1], self.faces_hand[h:h + 1]) for h in range(self.hand_nb)]
        fit_palette = ["gold"] + ["grey"] * self.hand_nb
        truth_palette = ["green"] + ["blue"] * self.hand_nb
        for suffix, meshes, palette in (("", scene, fit_palette), ("_gt", scene, truth_palette),
                                        ("_with_gt", scene + scene, fit_palette + truth_palette)):
            f, tex = get_faces_and_textures([v for v, _ in meshes], [t for _, t in meshes], color_names=palette)
            setattr(self, "faces" + suffix, f.repeat(batch_size, 1, 1))
            setattr(self, "textures" + suffix, tex.repeat(batch_size, 1, 1, 1, 1, 1))

    def render_limem(self, renderer, verts, faces, textures, K, max_in_batch=5):
        """:510-544: render in chunks, -> images (N,S,S,3) float in [0,1] (numpy), masks (N,S,S) bool."""
        sample_nb = verts.shape[0]
        assert verts.dim() == 3 and verts.shape[2] == 3
        assert tuple(faces.shape[::2]) == (sample_nb, 3) and tuple(textures.shape[2:]) == (1, 1, 1, 3)
        assert tuple(K.shape) == (sample_nb, 3, 3)
        chunk_nb = (sample_nb + 1) // min(max_in_batch, sample_nb) if max_in_batch is not None else 1
        all_images, all_masks = [], []
        for vert, face, tex, camintr in zip(verts.chunk(chunk_nb, 0), faces.chunk(chunk_nb, 0),
                                            textures.chunk(chunk_nb, 0), K.chunk(chunk_nb, 0)):
            chunk_images, _, chunk_masks = renderer.render(vertices=vert.contiguous(), faces=face.contiguous(),
                                                           textures=tex.contiguous(), K=camintr.contiguous())
            all_images.append(np.clip(chunk_images.cpu().numpy().transpose(0, 2, 3, 1), 0, 1))
            all_masks.append(chunk_masks.cpu().numpy().astype(bool))
        return np.concatenate(all_images), np.concatenate(all_masks)

    def _viz_K(self, renderer, n):
        K = renderer.K
        return (K.repeat(n, 1, 1) if K.shape[0] == 1 else K)[:n]

    def render(self, renderer, rotate=False, viz_len=10, max_in_batch=None):
        """:546-562."""
        with torch.no_grad():
            verts_object = self.get_verts_object()[0]
            verts_hands = self.get_verts_hand()[0]
            verts_hands = [verts_hands[i::self.hand_nb] for i in range(self.hand_nb)]
            verts_combined = combine_verts([verts_object] + verts_hands)
            if rotate:
                verts_combined = trans3d.rot_points(verts_combined)
            n = min(viz_len, verts_combined.shape[0])
            return self.render_limem(renderer, verts_combined[:viz_len], self.faces[:viz_len], self.textures[:viz_len],
                                     K=self._viz_K(renderer, n), max_in_batch=max_in_batch)

    def render_gt(self, renderer, verts_hand_gt=None, verts_object_gt=None, rotate=False, viz_len=10, max_in_batch=None):
        """:564-581."""
        with torch.no_grad():
            verts_combined = combine_verts([verts_object_gt, verts_hand_gt])
            if rotate:
                verts_combined = trans3d.rot_points(verts_combined)
            n = min(viz_len, verts_combined.shape[0])
            return self.render_limem(renderer, verts_combined[:viz_len], self.faces[:viz_len], self.textures_gt[:viz_len],
                                     K=self._viz_K(renderer, n), max_in_batch=max_in_batch)

    def render_with_gt(self, renderer, verts_hand_gt=None, verts_object_gt=None, rotate=False, viz_len=10, init=False,
                       max_in_batch=None):
        """:583-613."""
        with torch.no_grad():
            if init:
                verts_object_pred, verts_hands = self.verts_object_init, self.verts_hand_init
            else:
                verts_object_pred, verts_hands = self.get_verts_object()[0], self.get_verts_hand()[0]
            verts_hands_pred = [verts_hands[i::self.hand_nb] for i in range(self.hand_nb)]
            verts_list = [verts_object_pred] + verts_hands_pred + [verts_object_gt] + [v for v in verts_hand_gt]
            verts_combined = combine_verts(verts_list)
            if rotate:
                verts_combined = trans3d.rot_points(verts_combined)
            n = min(viz_len, verts_combined.shape[0])
            return self.render_limem(renderer, verts_combined[:viz_len], self.faces_with_gt[:viz_len],
                                     self.textures_with_gt[:viz_len], K=self._viz_K(renderer, n),
                                     max_in_batch=max_in_batch)

    def save_obj(self, fname):
        """:615-628: first scene of the clip as a Wavefront OBJ (combined object + hand mesh)."""
        with torch.no_grad():
            verts_combined = combine_verts([self.get_verts_object()[0], self.get_verts_hand()[0]])
        with open(fname, "w") as fp:
            for v in verts_combined[0].cpu().numpy():
                fp.write(f"v {v[0]:f} {v[1]:f} {v[2]:f}\n")
            for face in self.faces[0].cpu().numpy():
                fp.write(f"f {face[0] + 1:d} {face[1] + 1:d} {face[2] + 1:d}\n")

    # ------------------------------------------------------------------ vertices
    def get_verts_object(self):
        """reference homan.py:298-307."""
        return ops.rigid_transform(self.verts_object_og, self.rotations_object, self.translations_object,
                                   self.int_scales_object, abs_scale=True)

    def _mano_verts(self):
        if self._mano_cache is not None:
            return self._mano_cache
        h = self.hand_nb
        if h == 1:
            return self.mano_model.forward_pca(self.mano_pca_pose, rot=self.mano_rot, betas=self.mano_betas,
                                               side=self.hand_sides[0], trans=self.mano_trans)["verts"]
        # hands interleaved frame-major [h0_t0, h1_t0, h0_t1, ...] (homan.py:62-63): hand i is the strided slice i::h through
        # ITS side's layer (:343-358), re-interleaved
        per_hand = [self.mano_model.forward_pca(self.mano_pca_pose[i::h].contiguous(), rot=self.mano_rot[i::h].contiguous(),
                                                betas=self.mano_betas[i::h].contiguous(), side=side,
                                                trans=self.mano_trans[i::h].contiguous())["verts"]
                    for i, side in enumerate(self.hand_sides)]
        return torch.stack(per_hand, 1).reshape(-1, 778, 3)

    def get_verts_hand(self, detach_scale=False):
        """reference homan.py:341-382 (persp)."""
        verts_hand_og = self._mano_verts() if self.optimize_mano else self.verts_hand_og
        scale = self.int_scales_hand.detach() if detach_scale else self.int_scales_hand
        if self.hand_proj_mode == "persp":
            return ops.rigid_transform(verts_hand_og, self.rotations_hand, self.translations_hand, scale,
                                       abs_scale=False)
        if self.hand_proj_mode == "ortho":
            # reference homan.py:364-371 -> utils/camera.py:59-105: no rotation, the translation follows from the weak camera
            # `cams_hand` (a Parameter with `optimize_ortho_cam`), vertices = s * (v + trans).  On the kernels' rigid
            # transform that is (s * v) @ I + (s * trans); the twin detaches the MESH only (scale and camera keep their gradient
            # there, unlike the perspective twin), hence the second call on the detached mesh.  Eager / graph loops only.
            h = len(self.hand_sides)          # (cams_hand holds B * hand_nb rows, frame-major: K once per hand)
            K = self.camintr if h == 1 else self.camintr.repeat_interleave(h, dim=0)
            trans = weakcam_persp_trans(self.cams_hand, K)
            ident = torch.eye(3, device=trans.device)[:, :2].expand(trans.shape[0], 3, 2).contiguous()
            st = scale.view(-1, 1, 1) * trans
            full = ops.rigid_transform(verts_hand_og, ident, st, scale, abs_scale=False)[0]
            twin = ops.rigid_transform(verts_hand_og.detach(), ident, st, scale, abs_scale=False)[0]
            return full, twin
        raise ValueError(f"Expected hand_proj_mode {self.hand_proj_mode} to be in [ortho|persp]")

    def get_joints_hand(self):
        """reference homan.py:309-339 (21 joints incl. finger tips, camera space); no gradient."""
        with torch.no_grad():
            h = self.hand_nb
            joints = torch.stack([self.mano_model.joints(self.mano_pca_pose[i::h].contiguous(), self.mano_rot[i::h].contiguous(),
                                                         self.mano_betas[i::h].contiguous(), side=side)
                                  for i, side in enumerate(self.hand_sides)], 1).reshape(-1, 16, 3)
            verts = self._mano_verts() - self.mano_trans.unsqueeze(1)
            tips = verts[:, [745, 317, 444, 556, 673]]
            full = torch.cat([joints, tips], 1)[:, [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8,
                                                    9, 20]]
            full = full + self.mano_trans.unsqueeze(1)
            return ops.rigid_transform(full.contiguous(), self.rotations_hand, self.translations_hand,
                                       self.int_scales_hand, abs_scale=False)

    # ------------------------------------------------------------------ losses
    def depth_contexts(self):
        """(object raster context, hand raster context, object instance mask u8, hand instance mask u8) of the ordinal depth
        term, at the full-image size (built once)."""
        if self._depth_state is None:
            size = int(self.image_size)
            masks_o, masks_h = self.masks_object, self.masks_human
            if tuple(masks_o.shape[1:]) != (size, size) or tuple(masks_h.shape[1:]) != (size, size):
                raise NotImplementedError("ordinal depth: instance masks must be (B, image_size, image_size), got "
                                          f"{tuple(masks_o.shape)} / image_size {size}")
            batch, dev = self.translations_object.shape[0], self.translations_object.device
            ctx_o = ops.SilhouetteContext(self.faces_object, self.verts_object_og.shape[1], batch, size, dev)
            ctx_h = ops.SilhouetteContext(self.faces_hand[:1].expand(batch, -1, -1), 778, batch, size, dev)
            self._depth_state = (ctx_o, ctx_h, (masks_o != 0).to(torch.uint8).contiguous(),
                                 (masks_h != 0).to(torch.uint8).contiguous())
        return self._depth_state

    def _depth_contexts_hands(self):
        """two hands per frame: (object context, [one raster context per hand], object mask u8, [per-hand masks u8]) - a context
        carries ONE render's intermediates to its backward, so every layer has its own"""
        if getattr(self, "_depth_state_hands", None) is None:
            size = int(self.image_size)
            if tuple(self.masks_object.shape[1:]) != (size, size) or tuple(self.masks_human.shape[1:]) != (size, size):
                raise NotImplementedError("ordinal depth: instance masks must be (B, image_size, image_size)")
            batch, dev = self.translations_object.shape[0], self.translations_object.device
            ctx_o = ops.SilhouetteContext(self.faces_object, self.verts_object_og.shape[1], batch, size, dev)
            ctx_hs = [ops.SilhouetteContext(self.faces_hand[h:h + 1].expand(batch, -1, -1), 778, batch, size, dev)
                      for h in range(self.hand_nb)]
            m_hs = [(self.masks_human[h::self.hand_nb] != 0).to(torch.uint8).contiguous() for h in range(self.hand_nb)]
            self._depth_state_hands = (ctx_o, ctx_hs, (self.masks_object != 0).to(torch.uint8).contiguous(), m_hs)
        return self._depth_state_hands

    def compute_ordinal_depth_loss(self, verts_object=None, verts_hand=None):
        """reference homan/homan.py:384-419: render object and hand depth at the full-image intrinsics, compare their
        ordering with the instance masks (lossutils.py:133-169); with two hands the three layers pair-wise."""
        if verts_object is None:
            verts_object, _ = self.get_verts_object()
        if verts_hand is None:
            verts_hand, _ = self.get_verts_hand()
        if self.hand_nb != 1:
            # layers [object, hand 0, hand 1] (homan.py:403-417: one render per hand over its strided rows), every pair of them
            ctx_o, ctx_hs, m_o, m_hs = self._depth_contexts_hands()
            sils, deps = [], []
            for verts, ctx in [(verts_object, ctx_o)] + [(verts_hand[h::self.hand_nb].contiguous(), ctx_hs[h])
                                                         for h in range(self.hand_nb)]:
                sil, dep = ops.depth_render(verts, self.camintr, ctx, 1.0)
                sils.append(sil)
                deps.append(dep)
            return {"loss_depth": ops.ordinal_depth_loss_layers(deps, sils, [m_o] + m_hs, self.reduce_ws)}
        ctx_o, ctx_h, m_o, m_h = self.depth_contexts()
        sil_o, dep_o = ops.depth_render(verts_object, self.camintr, ctx_o, 1.0)
        sil_h, dep_h = ops.depth_render(verts_hand, self.camintr, ctx_h, 1.0)
        return {"loss_depth": ops.ordinal_depth_loss(dep_o, dep_h, sil_o, sil_h, m_o, m_h, self.reduce_ws)}

    def forward(self, loss_weights=None):
        """reference homan.py:421-508: a loss whose weight is zero is not computed."""
        lw = loss_weights
        on = lambda key: lw is None or lw[key] > 0
        loss_dict, metric_dict = {}, {}
        if self.optimize_mano:      # MANO is evaluated once and shared by the reference's two get_verts_hand calls
            self._mano_cache = None
            self._mano_cache = self._mano_verts()
        try:
            verts_object, _ = self.get_verts_object()
            verts_hand, verts_hand_det = self.get_verts_hand()
            if self.int_scales_hand.requires_grad:
                verts_hand_det_scale, _ = self.get_verts_hand(detach_scale=True)
            else:       # the scale is a buffer: detaching it changes nothing
                verts_hand_det_scale = verts_hand
        finally:
            self._mano_cache = None
        want_pca, want_so, want_sh = on("lw_pca"), on("lw_scale_obj"), on("lw_scale_hand")
        if want_pca or want_so or want_sh:
            l_pca, l_so, l_sh = ops.priors(self.mano_pca_pose, self.int_scales_object, self.int_scale_object_mean,
                                           self.int_scales_hand, self.int_scale_hand_mean)
            if want_pca:
                loss_dict["loss_pca"] = l_pca
        if lw is None or lw["lw_smooth_hand"] > 0 or lw["lw_smooth_obj"] > 0:
            loss_dict.update(lossutils.compute_smooth_loss(verts_hand, verts_object, self.reduce_ws))
        if on("lw_collision"):
            loss_dict.update(lossutils.compute_collision_loss(verts_hand_det_scale, verts_object.detach(),
                                                              self.collision_ctx))
        nn_cache = None
        if on("lw_contact"):
            l_contact, nn_cache = lossutils.compute_contact_loss(verts_hand_det_scale, verts_object, self.reduce_ws)
            loss_dict.update(l_contact)
        if on("lw_v2d_hand"):
            l, m = self.losses.compute_verts2d_loss_hand(verts_hand, image_size=self.image_size,
                                                         min_hand_size=70 if self.optimize_object_scale else 1000)
            loss_dict.update(l)
            metric_dict.update(m)
        if on("lw_sil_obj"):
            l, m = self.losses.compute_sil_loss_object(verts_object, self.faces_object)
            loss_dict.update(l)
            metric_dict.update(m)
        if on("lw_inter"):
            inter_obj = verts_object.unsqueeze(1) if self.optimize_object_scale else verts_object.unsqueeze(1).detach()
            l, m = self.losses.compute_interaction_loss(verts_hand_det.view(-1, self.hand_nb, 778, 3), inter_obj,
                                                        nn=nn_cache)
            loss_dict.update(l)
            metric_dict.update(m)
        if want_so:
            loss_dict["loss_scale_obj"] = l_so
        if want_sh:
            loss_dict["loss_scale_hand"] = l_sh
        if (lw is None or lw["lw_depth"] > 0) and self.ordinal_depth:
            loss_dict.update(self.compute_ordinal_depth_loss(verts_object, verts_hand))
        elif lw is None or lw["lw_depth"] > 0:
            # reference homan.py:506-507 calls lossutils.compute_ordinal_depth_loss() without its arguments
            raise TypeError("compute_ordinal_depth_loss() missing 3 required positional arguments: "
                            "'masks', 'silhouettes', and 'depths'")
        return loss_dict, metric_dict
