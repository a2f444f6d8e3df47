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
