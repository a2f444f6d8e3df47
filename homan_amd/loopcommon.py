"""What the optimisation loops share (reference homan/jointopt.py:55-151): collation of the per-frame input dicts, the three
name-selected Adam groups, the fused Adam launch, the device-side loss log, the model builder."""
import ctypes
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import lib as _lib
from .homan import HOMan


def _tensorify(x):
    if isinstance(x, torch.Tensor):
        return x
    a = np.asarray(x)
    return torch.from_numpy(a.astype(np.float32) if a.dtype.kind == "f" else a)


def collate_inputs(person_parameters, object_parameters, objvertices, objfaces):
    """Per-frame dicts -> HOMan keyword arguments (reference jointopt.py:52-91)."""
    cat = torch.cat
    pp, op = person_parameters, object_parameters
    return dict(
        hand_sides=pp[0]["hand_side"],
        translations_object=cat([o["translations"] for o in op]),
        rotations_object=cat([o["rotations"] for o in op]),
        verts_object_og=_tensorify(objvertices),
        faces_object=_tensorify(objfaces),
        target_masks_object=cat([o["target_masks"] for o in op]),
        target_masks_hand=cat([p["target_masks"] for p in pp]),
        verts_hand_og=cat([p["verts"] for p in pp]),
        ref_verts2d_hand=cat([p["verts2d"] for p in pp]),
        mano_trans=cat([p["mano_trans"] for p in pp]),
        mano_rot=cat([p["mano_rot"] for p in pp]),
        mano_pca_pose=cat([p["mano_pca_pose"] for p in pp]),
        mano_betas=cat([p["mano_betas"] for p in pp]),
        translations_hand=cat([p["translations"] for p in pp]),
        rotations_hand=cat([p["rotations"] for p in pp]),
        faces_hand=pp[0]["faces"],
        masks_object=cat([o["full_mask"].unsqueeze(0) for o in op]),
        masks_hand=cat([p["masks"] for p in pp]),
        cams_hand=cat([p["cams"] for p in pp]),
        camintr_rois_object=cat([o["K_roi"][:, 0] for o in op]),
        camintr_rois_hand=cat([p["K_roi"] for p in pp]),
    )


def parameter_groups(model, lr):
    """The three Adam groups of reference jointopt.py:128-151 (selected by parameter-name substring)."""
    rigid = [v for k, v in model.named_parameters() if "mano" not in k and "rotation" not in k]
    rotation = [v for k, v in model.named_parameters() if ("rotation" in k) and ("mano" not in k)]
    return [{"params": rigid, "lr": lr},
            {"params": [model.mano_pca_pose, model.mano_betas], "lr": lr * 10},
            {"params": rotation, "lr": lr * 10}]


class HmAdam:
    """Fused multi-tensor Adam on device (csrc/adam.hip): same arithmetic as torch's single-tensor Adam with
    betas=(0.9,0.999), eps=1e-8, one launch for all tensors, device-side step counter, gradients zeroed in the
    same launch.  Parameters whose .grad is None are skipped, like torch.optim.Adam does."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8):
        self.betas, self.eps = betas, eps
        self.items = []
        for g in groups:
            for p in g["params"]:
                if isinstance(p, torch.nn.Parameter) and p.requires_grad and p.grad is not None:
                    self.items.append((p, float(g["lr"])))
        assert self.items, "run one backward before building HmAdam (static gradient buffers)"
        dev = self.items[0][0].device
        self.state = [(torch.zeros_like(p), torch.zeros_like(p)) for p, _ in self.items]
        self.step_t = torch.zeros(2, dtype=torch.int32, device=dev)     # {steps done, ticket word of k_adam}
        slot = np.zeros(len(self.items), dtype=[("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"),
                                                 ("lr", "f4"), ("pad", "i4")])
        assert slot.itemsize == _lib.lib().hm_adam_slot_bytes()
        for i, ((p, lr), (m, v)) in enumerate(zip(self.items, self.state)):
            assert p.is_contiguous() and p.grad.is_contiguous()
            slot[i] = (p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, 0)
        self.slots = torch.from_numpy(slot.view(np.uint8).copy()).to(dev)
        self.grads = [p.grad for p, _ in self.items]       # keep the static buffers alive
        self.blocks = max(1, min(64, (max(p.numel() for p, _ in self.items) + 255) // 256))

    def step(self, zero_grad=True, log=None):
        """log = (vals (C, n+1), weights (n), n, max_steps, log_buf, C): the log row of the step being taken is written by the
        same launch (hm_adam_step_log = hm_log_total_clips + hm_adam_step)."""
        for (p, _), g in zip(self.items, self.grads):
            assert p.grad is g, "gradient buffers must stay static (do not call zero_grad(set_to_none=True))"
        if log is not None:
            vals, weights, n, max_steps, log_buf, nclips = log
            _lib.check(_lib.lib().hm_adam_step_log(_lib.ptr(self.slots), len(self.items), _lib.ptr(self.step_t),
                                                   self.betas[0], self.betas[1], self.eps, int(zero_grad), self.blocks,
                                                   _lib.ptr(vals), _lib.ptr(weights), n, max_steps, _lib.ptr(log_buf),
                                                   nclips, _lib.stream()), "hm_adam_step_log")
            return
        _lib.check(_lib.lib().hm_adam_step(_lib.ptr(self.slots), len(self.items), _lib.ptr(self.step_t),
                                           self.betas[0], self.betas[1], self.eps, int(zero_grad), self.blocks,
                                           _lib.stream()), "hm_adam_step")


class _DeviceLog:
    """loss_evolution without host syncs: the scalars of one iteration are packed and a kernel writes them into
    row `step` (device-side counter) of a (max_steps, n) buffer that is read back once at the end."""

    def __init__(self, keys, max_steps, step_t):
        self.keys = list(keys)
        self.buf = torch.zeros(max_steps, len(self.keys), device=step_t.device)
        self.max_steps, self.step_t = max_steps, step_t

    def record(self, scalars):
        packed = torch.cat([scalars[k].detach().reshape(1) for k in self.keys])
        _lib.check(_lib.lib().hm_log_scalars(_lib.ptr(packed), len(self.keys), _lib.ptr(self.step_t),
                                             self.max_steps, _lib.ptr(self.buf), _lib.stream()), "hm_log_scalars")


def _weighted_total(loss_dict, loss_weights):
    """loss = sum_k loss_k * lw[k.replace('loss','lw')]   (reference jointopt.py:180-188), shape (1,)."""
    return sum(loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict)


def build_model(person_parameters, object_parameters, class_name="default", objvertices=None, objfaces=None,
                camintr=None, hand_proj_mode="persp", optimize_mano=False, optimize_mano_beta=True,
                optimize_object_scale=False, state_dict=None, image_size=640, mano_model=None, rend_size=256,
                sync_metrics=True, ordinal_depth=False):
    kw = collate_inputs(person_parameters, object_parameters, objvertices, objfaces)
    model = HOMan(camintr=camintr, class_name=class_name, int_scale_init=1, hand_proj_mode=hand_proj_mode,
                  optimize_mano=optimize_mano, optimize_mano_beta=optimize_mano_beta,
                  optimize_object_scale=optimize_object_scale, image_size=image_size, mano_model=mano_model,
                  rend_size=rend_size, sync_metrics=sync_metrics, ordinal_depth=ordinal_depth, **kw)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=False)
    return model
