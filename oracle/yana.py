"""CPU restatements of the `libyana` helpers on the hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED (third-party
hassony2/libyana @ HEAD, reference requirements.txt:20; not in /root/reference).
Reference call sites: homan/losses.py:147 (batch_proj2d), :192 (batch_mask_iou),
:220,:227 (distutils.batch_pairwise_dist), jointopt.py:52-53 (npt.tensorify).
"""
import numpy as np
import torch


def tensorify(array, device=None):
    if isinstance(array, torch.Tensor):
        t = array
    else:
        a = np.asarray(array)
        t = torch.from_numpy(a.astype(np.float32) if a.dtype.kind == "f" else a)
    return t if device is None else t.to(device)


def numpify(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def batch_proj2d(verts, camintr, camextr=None):
    hom = camintr.bmm(verts.transpose(1, 2)).transpose(1, 2)
    return hom[:, :, :2] / hom[:, :, 2:]


def batch_mask_iou(ref, pred, eps=1e-6):
    ref, pred = ref.float(), pred.float()
    inter = ref * pred
    union = (ref + pred).clamp(0, 1)
    return inter.sum(1).sum(1) / (union.sum(1).sum(1) + eps)


def batch_pairwise_dist(x, y, use_cuda=False):
    """Squared distances (B,N,M); same algebra as reference interactions/contactloss.py:60-79."""
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    rx = torch.diagonal(xx, dim1=1, dim2=2).unsqueeze(1).expand_as(zz.transpose(2, 1))
    ry = torch.diagonal(yy, dim1=1, dim2=2).unsqueeze(1).expand_as(zz)
    return rx.transpose(2, 1) + ry - 2 * zz


def get_K_crop_resize(K, boxes, crop_resize):
    """libyana.lib3d.kcrop.get_K_crop_resize as called at reference homan/pose_optimization.py:246-248 and
    homan/homan.py (ROI intrinsics): intrinsics of the crop `boxes` (xyxy, pixels) resized to `crop_resize`.
    Third-party (hassony2/libyana @ HEAD, adapted from cosypose), absent from /root/reference: PARITY UNPINNED; restated
    from the published cosypose routine (pixel-centre convention: a crop of width w keeps its centre at (w-1)/2)."""
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


def batch_weakcam2persptrans(cams, Ks, reference_depth=1):
    """Scaled-orthographic cameras [s, tx, ty] in PIXEL units -> the camera-space translation under which the pinhole camera
    `Ks` images an object near the origin the way the weak camera does.  Reference call site: homan/utils/camera.py:96-97
    (`camconvs.batch_weakcam2persptrans(orthocams_pixels, K_pixels, 1)`).  The function lives in libyana
    (`libyana.camutils.camconvs`), which is NOT in /root/reference: this is the first-order camera identity, restated from
    the model and not from its source - PARITY UNPINNED, non-default `--hand_proj_mode ortho` only.
        weak camera:  u = s * X + t              pinhole:  u = f * (X + T_xy) / (Z + T_z) + c  ~  (f / T_z) * X + f * T_xy / T_z + c
        => T_z = reference_depth * f / s,   T_xy = (t - c) * T_z / f_xy
    (f: the focal length along x for the depth, per axis for the offsets).  cams (B,3), Ks (B or 1,3,3) -> (B,3)."""
    s = cams[:, 0]
    fx, fy = Ks[:, 0, 0], Ks[:, 1, 1]
    tz = reference_depth * fx / s
    tx = (cams[:, 1] - Ks[:, 0, 2]) * tz / fx
    ty = (cams[:, 2] - Ks[:, 1, 2]) * tz / fy
    return torch.stack([tx, ty, tz.expand_as(tx)], 1)
