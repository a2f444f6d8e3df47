"""Image-space losses with the reference's `Losses` surface (reference homan/losses.py:52-242), on HIP kernels."""
import torch

from . import constants, ops


class Losses:
    def __init__(self, renderer, ref_mask_object, ref_verts2d_hand, keep_mask_object, ref_mask_hand, keep_mask_hand,
                 camintr_rois_object, camintr_rois_hand, camintr, class_name, inter_type="min", hand_nb=1,
                 faces_object=None, num_verts_object=None, rend_size=constants.REND_SIZE, reduce_ws=None,
                 sync_metrics=True):
        if inter_type not in ("centroid", "min"):
            raise ValueError(f"inter_type {inter_type} not in [centroid|min]")
        self.inter_type = inter_type
        self.ref_mask_object, self.keep_mask_object = ref_mask_object, keep_mask_object
        self.ref_mask_hand, self.keep_mask_hand = ref_mask_hand, keep_mask_hand
        self.ref_verts2d_hand = ref_verts2d_hand
        self.camintr_rois_object, self.camintr_rois_hand = camintr_rois_object, camintr_rois_hand
        self.camintr = camintr.clone()
        self.thresh = constants.INTERACTION_Z_THRESH
        self.expansion = constants.INTERACTION_BBOX_EXPANSION
        self.class_name, self.hand_nb = class_name, hand_nb
        self.interaction_map = constants.INTERACTION_MAPPING[class_name]
        self.rws = reduce_ws
        self.sync_metrics = sync_metrics
        dev = ref_mask_object.device
        B = ref_mask_object.shape[0]
        self.sil_ctx = ops.SilhouetteContext(faces_object, num_verts_object, B, rend_size, dev)
        self.keep_sum = keep_mask_object.sum().reshape(1)
        self.last_silhouettes = None

    def _metric(self, t):
        return t.item() if self.sync_metrics else t.detach()

    def compute_verts2d_loss_hand(self, verts, image_size=640, min_hand_size=70):
        """reference losses.py:141-164 (the min_hand_size branch is computed and discarded there: no effect)."""
        loss, dist = ops.v2d_loss(verts, self.camintr, self.ref_verts2d_hand, image_size, self.hand_nb, self.rws)
        return {"loss_v2d_hand": loss}, {"v2d_hand": self._metric(dist)}

    def compute_sil_loss_object(self, verts, faces=None):
        """reference losses.py:183-197."""
        loss, iou, sil = ops.silhouette_loss(verts, self.camintr_rois_object, self.keep_mask_object,
                                             self.ref_mask_object, self.keep_sum, self.sil_ctx)
        self.last_silhouettes = sil
        return {"loss_sil_obj": loss}, {"iou_object": self._metric(iou)}

    def compute_sil_loss_hand(self, verts, faces):
        """reference losses.py:166-181 - present but disabled upstream (its only call site, homan.py:477-481, is commented
        out, and the body rebinds `verts` inside its own loop, so it cannot run past the first frame).  Built as written
        for the intent: each hand rendered in its own ROI camera (`camintr_rois_hand[i]`), masked L2 against its target
        normalised by ITS keep-mask area (unlike the object term, which divides by the area of the whole clip), averaged
        over the hands.  verts (N,778,3) camera space, faces (N,Fh,3).  Differentiable (NMR pseudo-gradient)."""
        if getattr(self, "_sil_ctx_hand", None) is None:
            if faces.shape[0] != verts.shape[0]:          # (hand_nb,Fh,3): hand i of the clip uses faces[i % hand_nb]
                faces = faces.repeat(verts.shape[0] // faces.shape[0], 1, 1)
            self._sil_ctx_hand = ops.SilhouetteContext(faces, verts.shape[1], verts.shape[0], self.ref_mask_hand.shape[-1],
                                                       verts.device)
        rend = ops.silhouette_render(verts, self.camintr_rois_hand, self._sil_ctx_hand)
        image = self.keep_mask_hand * rend
        per_hand = ((image - self.ref_mask_hand) ** 2).sum(dim=(1, 2)) / self.keep_mask_hand.sum(dim=(1, 2))
        return {"loss_sil_hand": per_hand.sum().reshape(1) / len(verts)}

    def compute_interaction_loss(self, verts_hand_b, verts_object_b, nn=None):
        """reference losses.py:199-242: verts_hand_b (B, hand_nb, 778, 3), verts_object_b (B, 1, V, 3).  The loss adds the
        terms of the hands (:207-239); the metric is the largest, over the frames, of the distance from the object to the
        NEAREST hand (:225-241).  `nn`: nearest-vertex search(es) already made by the contact term."""
        if verts_object_b.shape[1] != 1:
            raise NotImplementedError("one object per frame")
        vo = verts_object_b[:, 0]
        hand_nb = verts_hand_b.shape[1]
        if hand_nb == 1 and self.inter_type == "centroid":
            vh = verts_hand_b[:, 0]
            loss = ops.inter_loss(vh, vo, self.camintr, self.rws, self.expansion, self.thresh)
            if nn is None:
                nn = ops.nearest_vertices(vh, vo, self.rws)
            return {"loss_inter": loss}, {"handobj_maxdist": self._metric(nn[2][0])}
        if hand_nb == 1 and nn is not None:
            nn = [nn]
        loss, per_frame = None, []
        for p in range(hand_nb):
            vh = verts_hand_b[:, p].contiguous()
            search = nn[p] if nn is not None else ops.nearest_vertices(vh, vo, self.rws)
            if self.inter_type == "centroid":
                term = ops.inter_loss(vh, vo, self.camintr, self.rws, self.expansion, self.thresh)
            else:
                # inter_type "min" (:219-221): the smallest squared vertex distance of every interacting frame - the search
                # names the closest pair, the term itself is formed on the two gathered vertices (differentiable in both)
                flags = ops.interaction_flags(vh, vo, self.camintr, self.rws, self.expansion, self.thresh)
                i_star = search[1].argmin(1)
                j_star = search[0].gather(1, i_star[:, None]).long()[:, 0]
                rows = torch.arange(vh.shape[0], device=vh.device)
                diff = vh[rows, i_star] - vo[rows, j_star]
                term = ((diff * diff).sum(1) * flags.float()).sum().reshape(1)
            loss = term if loss is None else loss + term
            per_frame.append(search[1].min(1)[0])
        maxdist = torch.stack(per_frame).min(0)[0].max().clamp_min(0).sqrt()
        return {"loss_inter": loss}, {"handobj_maxdist": self._metric(maxdist)}
