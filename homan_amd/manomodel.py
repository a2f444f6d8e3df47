"""MANO wrapper with the reference's surface (reference homan/manomodel.py:15-151), backed by the fused HIP LBS."""
import torch
from torch import nn

from . import ops
from .mano_assets import get_mano


class ManoModel(nn.Module):
    """ManoModel(mano_root, pca_comps=16): `forward_pca(pca_pose, rot, betas, side)` -> {"verts","joints"}.

    Only the right hand is built (every BASELINE configuration uses one right hand); a left-hand request
    raises ValueError like an unknown side does in the reference (manomodel.py:141).
    """

    def __init__(self, mano_root="extra_data/mano", pca_comps=16, batch_size=1, mano_model=None, device="cuda"):
        super().__init__()
        self.pca_comps = pca_comps
        self.model_np = get_mano(mano_root) if mano_model is None else mano_model
        self.ctx_mean = ops.ManoContext(self.model_np, device, num_pca_comps=pca_comps, flat_hand_mean=False)
        self.ctx_flat = ops.ManoContext(self.model_np, device, num_pca_comps=pca_comps, flat_hand_mean=True)
        self.rh_mean = torch.as_tensor(self.model_np["hand_mean"])

    def forward_pca(self, pca_pose=None, rot=None, betas=None, side="right", flat_hand_mean=False, trans=None):
        if side != "right":
            raise ValueError(f"{side} not in [right] (left hand not built in homan_amd)")
        flatten = pca_pose.dim() == 1
        if flatten:
            pca_pose, rot = pca_pose.unsqueeze(0), rot.unsqueeze(0)
            betas = betas.unsqueeze(0) if betas is not None else None
        if betas is None:
            betas = torch.zeros(pca_pose.shape[0], 10, device=pca_pose.device)
        mctx = self.ctx_flat if flat_hand_mean else self.ctx_mean
        verts = ops.mano_lbs(pca_pose, rot, betas, trans, mctx)
        out = {"verts": verts[0] if flatten else verts}
        return out

    def joints(self, pca_pose, rot, betas, trans=None, flat_hand_mean=False):
        mctx = self.ctx_flat if flat_hand_mean else self.ctx_mean
        return ops.mano_joints(pca_pose, rot, betas, trans, mctx)[1]

    @property
    def closed_faces(self):
        return self.model_np["closed_faces"]
