"""Frontal overlay + top-down renders of a fitted clip (reference homan/visualize.py:44-128), on the HIP rasteriser.

Only `visualize_hand_object` is mirrored - the function the optimisation loop calls (reference homan/jointopt.py:176-206).
`visualize_perspective` / `visualize_orthographic` (:9-41) wrap look-at renderers of the PHOSA code base that the
hand-object pipeline never calls.
"""
import numpy as np


def visualize_hand_object(model, images, verts_hand_gt=None, verts_object_gt=None, dist=3, viz_len=7, init=False,
                          gt_only=False, image_size=640, max_in_batch=2):
    """-> (frontal (n,h,w,3) uint8: renders pasted over the input images, top_down (n,S,S,3) uint8: the scene rotated
    about its centroid).  `dist` / `image_size` only parametrise a second renderer the reference builds and never uses
    (:86-101): the top-down view is rendered by `model.renderer` on rotated vertices (:102-125)."""
    if gt_only:
        rends, masks = model.render_gt(model.renderer, verts_hand_gt=verts_hand_gt, verts_object_gt=verts_object_gt,
                                       viz_len=viz_len, max_in_batch=max_in_batch)
    elif verts_hand_gt is None:
        rends, masks = model.render(model.renderer, viz_len=viz_len, max_in_batch=max_in_batch)
    else:
        rends, masks = model.render_with_gt(model.renderer, verts_hand_gt=verts_hand_gt, verts_object_gt=verts_object_gt,
                                            viz_len=viz_len, init=init, max_in_batch=max_in_batch)
    new_images = []
    for image, rend, mask in zip(images, rends, masks):
        if image.max() > 1:
            image = image / 255.0
        h, w, _ = image.shape
        L = max(h, w)
        new_image = np.pad(image.copy(), ((0, L - h), (0, L - w), (0, 0)))
        new_image[mask] = rend[mask]
        new_images.append((new_image[:h, :w] * 255).astype(np.uint8))
    if verts_hand_gt is None:
        top_down, _ = model.render(model.renderer, rotate=True, viz_len=viz_len, max_in_batch=max_in_batch)
    elif gt_only:
        top_down, _ = model.render_gt(model.renderer, verts_hand_gt=verts_hand_gt, verts_object_gt=verts_object_gt,
                                      viz_len=viz_len, rotate=True, max_in_batch=max_in_batch)
    else:
        top_down, _ = model.render_with_gt(model.renderer, verts_hand_gt=verts_hand_gt, verts_object_gt=verts_object_gt,
                                           rotate=True, viz_len=viz_len, init=init, max_in_batch=max_in_batch)
    return np.stack(new_images), (top_down * 255).astype(np.uint8)
