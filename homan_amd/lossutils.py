"""3-D space losses with the reference's function surface (reference homan/lossutils.py), on HIP kernels."""
import torch

from . import constants, ops


def compute_smooth_loss(verts_hand, verts_obj, rws):
    """reference lossutils.py:18-36."""
    hand_nb = verts_hand.shape[0] // verts_obj.shape[0]
    return {"loss_smooth_obj": ops.smooth_loss(verts_obj, 1, rws),
            "loss_smooth_hand": ops.smooth_loss(verts_hand, hand_nb, rws)}


def compute_collision_loss(verts_hand, verts_object, cctx):
    """reference lossutils.py:43-64 (collision_mode='sdf').  `cctx`: one CollisionContext (one hand), or the three of the
    two-hand scene [hand 0, hand 1, object] - (hand 0, hand 1), (hand 0, object), (hand 1, object) - whose two-mesh
    losses add up to the reference's sum over the six ordered pairs of meshes (scenesdf.py:131-146)."""
    if not isinstance(cctx, (tuple, list)):
        return {"loss_collision": ops.collision_loss(verts_hand, verts_object, cctx, constants.SDF_SCALE_FACTOR)}
    h0, h1 = verts_hand[0::2].contiguous(), verts_hand[1::2].contiguous()          # stride 2, lossutils.py:57-59
    loss = ops.collision_loss(h0, h1, cctx[0], constants.SDF_SCALE_FACTOR)
    loss = loss + ops.collision_loss(h0, verts_object, cctx[1], constants.SDF_SCALE_FACTOR)
    loss = loss + ops.collision_loss(h1, verts_object, cctx[2], constants.SDF_SCALE_FACTOR)
    return {"loss_collision": loss}


def compute_contact_loss(verts_hand_b, verts_object_b, rws, nn=None):
    """reference lossutils.py:112-130 -> interactions/contactloss.py:149-309 (as executed: SURVEY appendix B.1).  Several
    hands (:116-127): the per-hand terms (hand i = rows i::hand_nb) averaged.  -> ({loss}, nearest-vertex search of the
    hand(s): one tuple, or a list of one per hand)."""
    hand_nb = verts_hand_b.shape[0] // verts_object_b.shape[0]
    if hand_nb == 1:
        if nn is None:
            nn = ops.nearest_vertices(verts_hand_b, verts_object_b, rws)
        return {"loss_contact": ops.contact_loss(verts_hand_b, verts_object_b, nn[0], rws, constants.COLLISION_THRESH)}, nn
    nns, terms = [], []
    for i in range(hand_nb):
        vh = verts_hand_b[i::hand_nb].contiguous()
        nns.append(ops.nearest_vertices(vh, verts_object_b, rws) if nn is None else nn[i])
        terms.append(ops.contact_loss(vh, verts_object_b, nns[-1][0], rws, constants.COLLISION_THRESH))
    return {"loss_contact": torch.stack([t.reshape(()) for t in terms]).mean()}, nns      # 0-d, like the reference's (:126-127)
