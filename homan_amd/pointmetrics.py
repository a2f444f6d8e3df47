"""Interaction metrics on the HIP SDF kernels.

Counterpart of `get_inter_metrics` (reference homan/eval/pointmetrics.py:102-124, called by fit_vid_dataset.py:488-493):
the one evaluation metric that a hot-path kernel computes - how deep the hand reaches into the object, read off the
object's signed-distance grid.  The chamfer / ADD-S / aligned-vertex metrics of the same reference file (:18-99) are
pytorch3d / cKDTree calls on CPU-side evaluation data and stay out of scope.
"""
import torch

from . import constants, ops


def _one_hand_mesh_per_scene(verts_person, faces_person, scenes):
    """(scenes * hands, Vh, 3) vertices ordered scene-major and (hands, Fh, 3) faces -> ONE mesh per scene: the hands'
    vertices laid end to end and their faces re-indexed into that concatenation.  (The reference offsets the second hand's
    faces by the length of the ALREADY merged vertex array, pointmetrics.py:104-110, i.e. past its end; the offset that
    indexes the second hand's block - one hand's vertex count - is used here.)"""
    hands, per_hand = verts_person.shape[0] // scenes, verts_person.shape[1]
    if hands > 3:
        raise ValueError(f"Invalid hand nb {hands}")
    faces_person = torch.as_tensor(faces_person)
    if hands == 1:
        return verts_person, faces_person[0]
    merged = verts_person.reshape(scenes, hands * per_hand, 3)
    faces = torch.cat([faces_person[h % faces_person.shape[0]] + h * per_hand for h in range(hands)], 0)
    return merged, faces


def get_inter_metrics(verts_person, verts_object, faces_person, faces_object):
    """verts_person (B*hand_nb, Vh, 3), verts_object (B, Vo, 3), faces_person (hand_nb, Fh, 3) closed hand faces,
    faces_object (>=1, Fo, 3)  ->  {"pen_depths": [B floats], "has_contact": [B bools]}.
    pen_depth = the largest value the object's clamped SDF takes at a hand vertex, in world units
    (`sdf_meta["dist_values"][(1, 0)]` of reference interactions/scenesdf.py:141-146); contact = any vertex inside."""
    scenes = verts_object.shape[0]
    hand_verts, hand_faces = _one_hand_mesh_per_scene(verts_person, faces_person, scenes)
    cctx = ops.CollisionContext(hand_faces.cpu().numpy(), torch.as_tensor(faces_object)[0], scenes, hand_verts.shape[1],
                                verts_object.shape[1], verts_object.device)
    depth_in_object = ops.collision_dist_values(hand_verts.contiguous(), verts_object, cctx, constants.SDF_SCALE_FACTOR)[(1, 0)]
    deepest = depth_in_object.amax(dim=1)
    return {"pen_depths": deepest.cpu().numpy().tolist(), "has_contact": (deepest > 0).cpu().numpy().tolist()}
