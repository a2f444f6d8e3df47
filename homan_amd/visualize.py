"""Frontal overlay + top-down renders of a fitted clip, on the HIP rasteriser.

Counterpart of `visualize_hand_object` (reference homan/visualize.py:44-128), the one visualisation function the
optimisation loop calls (reference homan/jointopt.py:159-177).  Same signature and return value:

    frontal  (n, h, w, 3) uint8   the scene rendered by `model.renderer` pasted over the input images
    top_down (n, S, S, 3) uint8   the same scene rotated about its centroid (`rotate=True` of the model's render methods)

Three scene variants exist upstream - prediction, ground truth only, prediction + ground truth - each rendered twice
(frontal, rotated).  Here one dispatcher picks the variant and both views go through it.  `dist` / `image_size` are
accepted for signature compatibility only: upstream they parametrise a second look-at renderer that is built and never
used (:86-101).  `visualize_perspective` / `visualize_orthographic` (:9-41) belong to the PHOSA code base and are never
called by the hand-object pipeline.
"""
import numpy as np


def _scene_render(model, variant, rotate, verts_hand_gt, verts_object_gt, viz_len, init, max_in_batch):
    """images (n,S,S,3) float in [0,1] and coverage masks (n,S,S) bool of one scene variant, frontal or rotated"""
    common = dict(viz_len=viz_len, rotate=rotate, max_in_batch=max_in_batch)
    if variant == "pred":
        return model.render(model.renderer, **common)
    gt = dict(verts_hand_gt=verts_hand_gt, verts_object_gt=verts_object_gt)
    if variant == "gt":
        return model.render_gt(model.renderer, **gt, **common)
    return model.render_with_gt(model.renderer, init=init, **gt, **common)


def _paste(image, rend, mask):
    """render over photo where the scene covers it; photo (h,w,3) in [0,255] or [0,1], render (S,S,3) in [0,1] anchored at
    the top-left corner (a non-square photo sees the part of the square render that overlaps it)"""
    photo = np.asarray(image, dtype=np.float64)
    if photo.max() > 1:
        photo = photo / 255.0
    h, w = photo.shape[:2]
    hh, ww = min(h, rend.shape[0]), min(w, rend.shape[1])
    out = photo.copy()
    region = out[:hh, :ww]
    covered = mask[:hh, :ww]
    region[covered] = rend[:hh, :ww][covered]
    return (out * 255).astype(np.uint8)


def visualize_hand_object(model, images, verts_hand_gt=None, verts_object_gt=None, dist=3, viz_len=7, init=False,
                          gt_only=False, image_size=640, max_in_batch=2):
    variant = "gt" if gt_only else ("pred" if verts_hand_gt is None else "pred+gt")
    args = (verts_hand_gt, verts_object_gt, viz_len, init, max_in_batch)
    rends, masks = _scene_render(model, variant, False, *args)
    frontal = np.stack([_paste(img, rend, mask) for img, rend, mask in zip(images, rends, masks)])
    # upstream decides the rotated view's variant in a different order (no ground-truth hand -> prediction, even with
    # gt_only set): kept
    top_variant = "pred" if verts_hand_gt is None else variant
    top_down, _ = _scene_render(model, top_variant, True, *args)
    return frontal, (top_down * 255).astype(np.uint8)
