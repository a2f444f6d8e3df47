"""Interaction metrics on the HIP SDF kernels (reference homan/eval/pointmetrics.py:102-124).

Only `get_inter_metrics` is mirrored: it is the one evaluation metric computed by a hot-path kernel (the scene SDF).  The
chamfer / ADD-S / aligned-vertex metrics of the same reference file (:18-99) are pytorch3d / cKDTree calls on CPU-side
evaluation data and stay out of scope.
"""
import torch

from . import constants, ops


def get_inter_metrics(verts_person, verts_object, faces_person, faces_object):
    """reference pointmetrics.py:102-124.  verts_person (B*hand_nb, Vh, 3), verts_object (B, Vo, 3), faces_person
    (hand_nb, Fh, 3) closed hand faces, faces_object (>=1, Fo, 3).  Returns {"pen_depths": [B floats] maximal depth of a
    hand vertex inside the object, "has_contact": [B bools] pen_depth > 0}.  Two hands are merged into one mesh per scene
    exactly as the reference does (:104-110)."""
    hand_nb = verts_person.shape[0] // verts_object.shape[0]
    faces_person = torch.as_tensor(faces_person)
    if hand_nb == 2:
        verts_person = verts_person.view(-1, hand_nb, verts_person.shape[1], 3).view(verts_object.shape[0], -1, 3)
        faces_person = torch.cat([faces_person[0], faces_person[1] + verts_person.shape[1]], 0).unsqueeze(0)
    elif hand_nb > 3:
        raise ValueError(f"Invalid hand nb {hand_nb}")
    B = verts_object.shape[0]
    cctx = ops.CollisionContext(faces_person[0].cpu().numpy(), torch.as_tensor(faces_object)[0], B,
                                verts_person.shape[1], verts_object.shape[1], verts_object.device)
    dist_values = ops.collision_dist_values(verts_person, verts_object, cctx, constants.SDF_SCALE_FACTOR)
    max_depths = dist_values[(1, 0)].max(1)[0]          # penetration of the hand into the object
    has_contact = max_depths > 0                        # at least one vertex inside
    return {"pen_depths": max_depths.cpu().numpy().tolist(), "has_contact": has_contact.cpu().numpy().tolist()}
