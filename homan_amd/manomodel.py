"""MANO wrapper with the reference's surface (reference homan/manomodel.py:15-151), backed by the fused HIP LBS."""
import torch
from torch import nn

import numpy as np

from . import ops
from .mano_assets import hand_models


class ManoModel(nn.Module):
    """ManoModel(mano_root, pca_comps=16): `forward_pca(pca_pose, rot, betas, side)` -> {"verts"}.

    Right and left hands (reference manomodel.py:19-80 loads MANO_RIGHT.pkl / MANO_LEFT.pkl; here `mano_model` may be a
    {"right", "left"} dictionary of model dicts, or one right-hand model whose mirror image serves as the left hand).
    The left path of the reference (:124-140) flips the sign of the y / z components of every joint's axis-angle after the
    PCA expansion and before the mean pose is added: folded into the left context's PCA basis, so one kernel serves both.
    An unknown side raises ValueError like in the reference (:141).
    """

    def __init__(self, mano_root="extra_data/mano", pca_comps=16, batch_size=1, mano_model=None, device="cuda"):
        super().__init__()
        self.pca_comps = pca_comps
        self.models = hand_models(mano_model, mano_root)
        self.model_np = self.models["right"]
        self.device = device
        self.ctx_mean = ops.ManoContext(self.model_np, device, num_pca_comps=pca_comps, flat_hand_mean=False)
        self.ctx_flat = ops.ManoContext(self.model_np, device, num_pca_comps=pca_comps, flat_hand_mean=True)
        self._left = {}
        self.rh_mean = torch.as_tensor(self.model_np["hand_mean"])

    def _left_ctx(self, flat):
        if flat not in self._left:
            m = dict(self.models["left"])
            comps = np.array(m["hand_components"], np.float32, copy=True)
            comps[:, 1::3] *= -1.0           # hand_pose[:, 1::3] *= -1 ; hand_pose[:, 2::3] *= -1  (manomodel.py:131-132)
            comps[:, 2::3] *= -1.0
            m["hand_components"] = comps
            self._left[flat] = ops.ManoContext(m, self.device, num_pca_comps=self.pca_comps, flat_hand_mean=flat)
        return self._left[flat]

    def forward_pca(self, pca_pose=None, rot=None, betas=None, side="right", flat_hand_mean=False, trans=None):
        if side not in ("right", "left"):
            raise ValueError(f"{side} not in [left|right]")
        flatten = pca_pose.dim() == 1
        if flatten:
            pca_pose, rot = pca_pose.unsqueeze(0), rot.unsqueeze(0)
            betas = betas.unsqueeze(0) if betas is not None else None
        if betas is None:
            betas = torch.zeros(pca_pose.shape[0], 10, device=pca_pose.device)
        if side == "right":
            mctx = self.ctx_flat if flat_hand_mean else self.ctx_mean
        else:
            mctx = self._left_ctx(bool(flat_hand_mean))
        verts = ops.mano_lbs(pca_pose, rot, betas, trans, mctx)
        out = {"verts": verts[0] if flatten else verts}
        return out

    def joints(self, pca_pose, rot, betas, trans=None, flat_hand_mean=False, side="right"):
        mctx = (self.ctx_flat if flat_hand_mean else self.ctx_mean) if side == "right" else self._left_ctx(bool(flat_hand_mean))
        return ops.mano_joints(pca_pose, rot, betas, trans, mctx)[1]

    @property
    def closed_faces(self):
        return self.model_np["closed_faces"]
