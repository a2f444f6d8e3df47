"""Clip batches: C independent clips optimised as ONE launch per kernel (BASELINE cfg4 / cfg5, SURVEY 8e "Partitioning").

The reference optimises clips one after the other, one `HOMan` + one Adam per clip (homan/jointopt.py:92-151; its only
sharding hook is the strided sample selection of fit_vid_dataset.py:54-55,190).  Clips share nothing, but the frames of
one clip are coupled by the smoothness term and the per-clip normalisers, so the batch axis is "whole clips laid end to
end": `ClipBatch` concatenates the per-frame Parameters / buffers of C `HOMan` models of identical shapes along the
frame axis, stacks the per-clip scalars (intrinsic scales, sum(keep)) into (C,) arrays, and re-points every model's
Parameters to views of the batched storage, so after a batched optimisation each model holds its own result
(`model.state_dict()`, `get_verts_*` ... behave as if it had been optimised alone).

The kernels see the batch through the `*_clips` entry points of include/homan_amd.h: per clip, every sum is formed by
the clip's own workgroups in the order of a single-clip launch, so a batched step is bit-identical to C single steps.
"""
import torch
from torch import nn

from . import lib as _lib
from . import ops

# Parameters / buffers concatenated along the frame axis (leading dimension = frames of the clip)
_PER_FRAME = ["translations_object", "rotations_object", "translations_hand", "rotations_hand", "cams_hand",
              "mano_pca_pose", "mano_rot", "mano_trans", "mano_betas", "verts_object_og", "verts_hand_og", "ref_verts2d_hand",
              "ref_mask_object", "keep_mask_object", "camintr_rois_object", "camintr"]
# (1,) per model -> (C,) per batch
_PER_CLIP = ["int_scales_object", "int_scales_hand", "int_scale_object_mean", "int_scale_hand_mean"]


class ClipReduceWorkspace:
    """One zero-initialised reduce-workspace slice (partials + self-resetting ticket) per clip, back to back."""

    def __init__(self, device, clips):
        self.buf = torch.zeros(clips * _lib.lib().hm_reduce_workspace_bytes(), dtype=torch.uint8, device=device)


class ClipBatch(nn.Module):
    def __init__(self, models):
        super().__init__()
        models = list(models)
        assert models, "ClipBatch needs at least one model"
        m0 = models[0]
        self.models, self.C = models, len(models)
        self.clip_len = m0.translations_object.shape[0]
        self.B = self.C * self.clip_len
        for k in ("optimize_mano", "optimize_object_scale", "hand_proj_mode", "image_size", "hand_nb"):
            vals = {getattr(m, k) for m in models}
            assert len(vals) == 1, f"clips of one batch must agree on {k}: {vals}"
            setattr(self, k, getattr(m0, k))
        self.ordinal_depth = any(getattr(m, "ordinal_depth", False) for m in models)
        for m in models[1:]:
            for k in _PER_FRAME:
                if hasattr(m0, k):
                    assert getattr(m, k).shape == getattr(m0, k).shape, f"clips of one batch must share the shape of {k}"
            assert m.int_scales_hand.requires_grad == m0.int_scales_hand.requires_grad
            assert torch.equal(m.faces_object[0], m0.faces_object[0]), "clips of one batch share the object topology"
        object.__setattr__(self, "mano_model", m0.mano_model)     # constant model data, shared (not a submodule)
        dev = m0.translations_object.device
        if self.C == 1:
            # a single clip IS its own batch: no copies, the stepper works on the model's tensors
            for k in _PER_FRAME + _PER_CLIP:
                if hasattr(m0, k):
                    self._adopt(k, getattr(m0, k))
            self.sil_ctx, self.keep_sum = m0.losses.sil_ctx, m0.losses.keep_sum
            self.collision_ctx = m0.collision_ctx
        else:
            for k in _PER_FRAME + _PER_CLIP:
                if not hasattr(m0, k):
                    continue
                parts = [getattr(m, k) for m in models]
                cat = torch.cat([p.detach() for p in parts]).contiguous()
                if isinstance(parts[0], nn.Parameter):
                    cat = nn.Parameter(cat, requires_grad=parts[0].requires_grad)
                self._adopt(k, cat)
                n = parts[0].shape[0]
                if isinstance(parts[0], nn.Parameter):      # each model keeps seeing (and owning a view of) its clip
                    for i, p in enumerate(parts):
                        p.data = cat.data[i * n:(i + 1) * n]
            faces = m0.faces_object[:1].expand(self.B, -1, -1)
            self.sil_ctx = ops.SilhouetteContext(faces, m0.verts_object_og.shape[1], self.B, m0.losses.sil_ctx.size, dev)
            self.keep_sum = torch.cat([m.losses.keep_sum for m in models]).contiguous()
            self.collision_ctx = ops.CollisionContext(m0.mano_model.closed_faces, m0.faces_object[0], self.B, 778,
                                                      m0.verts_object_og.shape[1], dev)
        self.reduce_ws = ClipReduceWorkspace(dev, self.C)

    def _adopt(self, name, t):
        if isinstance(t, nn.Parameter):
            self.register_parameter(name, t)
        else:
            self.register_buffer(name, t, persistent=False)

    def clip_slice(self, t, c):
        n = t.shape[0] // self.C
        return t[c * n:(c + 1) * n]
