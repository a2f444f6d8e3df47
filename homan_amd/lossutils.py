"""3-D space losses with the reference's function surface (reference homan/lossutils.py), on HIP kernels."""
from . import constants, ops


def compute_smooth_loss(verts_hand, verts_obj, rws):
    """reference lossutils.py:18-36."""
    hand_nb = verts_hand.shape[0] // verts_obj.shape[0]
    return {"loss_smooth_obj": ops.smooth_loss(verts_obj, 1, rws),
            "loss_smooth_hand": ops.smooth_loss(verts_hand, hand_nb, rws)}


def compute_collision_loss(verts_hand, verts_object, cctx):
    """reference lossutils.py:43-64 (collision_mode='sdf', one hand)."""
    return {"loss_collision": ops.collision_loss(verts_hand, verts_object, cctx, constants.SDF_SCALE_FACTOR)}


def compute_contact_loss(verts_hand_b, verts_object_b, rws, nn=None):
    """reference lossutils.py:112-130 -> interactions/contactloss.py:149-309 (as executed: SURVEY appendix B.1)."""
    if nn is None:
        nn = ops.nearest_vertices(verts_hand_b, verts_object_b, rws)
    return {"loss_contact": ops.contact_loss(verts_hand_b, verts_object_b, nn[0], rws, constants.COLLISION_THRESH)}, nn
