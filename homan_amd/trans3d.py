"""`libyana.lib3d.trans3d.rot_points` as called by reference homan/homan.py:555,574,606 (top-down visualisation view).
libyana is an un-pinned git dependency that is not in /root/reference: this is a restatement of the published helper
from recollection (UNVERIFIED) - rotate every scene's points about their own centroid by a fixed axis-angle vector
(default (0, 1, 1), Rodrigues)."""
import torch


def _rodrigues(axisang):
    aa = torch.as_tensor(axisang, dtype=torch.float32)
    angle = torch.norm(aa + 1e-8)
    k = aa / angle
    K = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=torch.float32)
    return torch.eye(3) + torch.sin(angle) * K + (1 - torch.cos(angle)) * (K @ K)


def rot_points(points, centers=None, axisang=(0, 1, 1)):
    if points.dim() != 3 or points.shape[2] != 3:
        raise ValueError(f"Expected batch of vertices in format (batch_size, vert_nb, 3) but got {points.shape}")
    if centers is None:
        centers = points.mean(1)
    R = _rodrigues(axisang).to(points.device)
    centred = points - centers.unsqueeze(1)
    return torch.matmul(centred, R.t()) + centers.unsqueeze(1)
