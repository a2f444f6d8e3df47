"""Joint hand-object optimisation loop (reference homan/jointopt.py:22-201), MI355X-native.

`optimize_hand_object` keeps the reference signature and return value `(model, loss_evolution, imgs)`.  When the
input `images` are given, a frontal + top-down frame is rendered every `viz_step` iterations and `imgs` maps the
step to the saved file (reference :158-176); the video export (libyana np2vid, :193-200) is not built.

Execution modes (`mode=`, a homan_amd extension; default "auto" = "fused" when it covers the configuration, else "graph"):
  mode="eager"  the reference's loop verbatim: torch.optim.Adam over the three name-selected groups, per-key
                `.item()` logging every step (host-synchronous, like the reference).
  mode="fused"  the iteration as a fixed sequence of C-ABI kernel launches without the autograd tape (FusedStepper),
                captured in a hipGraph: ~27 launches instead of ~90.  The benchmark path.
  mode="graph"  the same iteration (zero-grad, forward, weighting, backward, Adam with the same arithmetic, logging)
                captured once into a hipGraph and replayed; losses / metrics are written to a device log by a kernel
                and read back once at the end (no host sync inside the loop).
"""
import ctypes
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import lib as _lib
from .homan import HOMan
from .fused import FusedStepper, _loop_streams, _morton_order  # noqa: F401  (the loops live in their own modules; this is the reference's module name)
from .loopcommon import (HmAdam, _DeviceLog, _tensorify, _weighted_total, build_model, collate_inputs,  # noqa: F401
                         parameter_groups)
from .shard import ClipFitter, ShardStepper, _input_signature, _shape_signature  # noqa: F401


class GraphStepper:
    """One optimisation iteration captured in a hipGraph (torch.cuda.CUDAGraph stream capture): forward, weighting,
    sync-free logging, backward, fused Adam (+ gradient zeroing).  `run(n)` replays it n times."""

    def __init__(self, model, loss_weights, lr, max_steps):
        self.model, self.lw, self.max_steps = model, dict(loss_weights), max_steps
        model.losses.sync_metrics = False
        model.sync_metrics = False
        params = [p for p in model.parameters() if p.requires_grad]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # one un-stepped backward tells which parameters receive gradients; give those static zero buffers
            self._fwd_bwd()
            for p in params:
                if p.grad is not None:
                    p.grad = torch.zeros_like(p)
            self.opt = HmAdam(parameter_groups(model, lr))
            for _ in range(2):               # warm-up (allocator, lazy init); no optimiser step is taken
                self._fwd_bwd()
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()
            loss_dict, metric_dict = model(loss_weights=self.lw)
            self.log = _DeviceLog(list(loss_dict) + list(metric_dict) + ["loss"], max_steps, self.opt.step_t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = _lib.new_graph()
        with torch.cuda.graph(self.graph, pool=_lib.autograd_pool()):
            loss_dict, metric_dict, total = self._fwd_bwd(record=True)
            self.opt.step()
        # Nothing allocated inside the capture stays referenced: captures share ONE memory pool (lib.autograd_pool), so a
        # tensor kept from this capture sits in a block an older stepper's replay reuses as a temporary - two live graph-mode
        # steppers replayed alternately would silently overwrite each other's kept tensors.  What a caller reads
        # lives outside the pool: Parameters, their .grad buffers and the log (allocated before the capture).
        del loss_dict, metric_dict, total
        model.losses.last_silhouettes = None
        model._mano_cache = None

    def _fwd_bwd(self, record=False):
        loss_dict, metric_dict = self.model(loss_weights=self.lw)
        total = _weighted_total(loss_dict, self.lw)
        if record:
            scalars = OrderedDict(list(loss_dict.items()) + list(metric_dict.items()) + [("loss", total)])
            self.log.record(scalars)
        total.sum().backward()
        return loss_dict, metric_dict, total

    def run(self, steps):
        for _ in range(steps):
            self.graph.replay()

    def reload(self, clip_inputs):
        """Another clip of the same shapes into the resident stepper (what FusedStepper.reload does for the fused loop): the
        clip's data copied in place into the model, everything that belongs to the FIT reset - Adam moments, step counter,
        gradients, log -, the captured graph reused as it is."""
        with torch.no_grad():
            self.model.load_clip(**clip_inputs)
            for st_m, st_v in self.opt.state:
                st_m.zero_()
                st_v.zero_()
            self.opt.step_t.zero_()
            self.log.buf.zero_()
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad.zero_()

    def loss_evolution(self, steps):
        torch.cuda.synchronize()
        host = self.log.buf[:steps].cpu().numpy()
        return {k: host[:, i].astype(np.float64).tolist() for i, k in enumerate(self.log.keys)}


def save_front_top(model, images, step, viz_folder, viz_len=7):
    """reference jointopt.py:159-176: frontal overlays over the top-down renders, frames side by side, halved in size
    (2x2 mean instead of cv2.resize), saved as <viz_folder>/<step:08d>.jpg.  -> path."""
    import os
    from PIL import Image
    from .visualize import visualize_hand_object
    with torch.no_grad():
        frontal, top_down = visualize_hand_object(model, images, dist=1, viz_len=viz_len)
    os.makedirs(viz_folder, exist_ok=True)
    frontal = np.concatenate([img for img in frontal], 1)
    top_down = np.concatenate([img for img in top_down], 1)
    front_top = np.concatenate([frontal, top_down[:frontal.shape[0], :frontal.shape[1]]], 0)
    h, w = front_top.shape[0] // 2 * 2, front_top.shape[1] // 2 * 2
    front_top = front_top[:h, :w].reshape(h // 2, 2, w // 2, 2, 3).astype(np.float32).mean((1, 3)).astype(np.uint8)
    path = os.path.join(viz_folder, f"{step:08d}.jpg")
    Image.fromarray(front_top).save(path)
    return path


def optimize_hand_object(person_parameters, object_parameters, class_name="default", objvertices=None, objfaces=None,
                         loss_weights=None, num_iterations=400, lr=1e-2, images=None, viz_step=10, viz_folder="tmp",
                         camintr=None, hand_proj_mode="persp", optimize_mano=False, optimize_mano_beta=True,
                         optimize_object_scale=False, state_dict=None, fps=24, viz_len=7, image_size=640,
                         # homan_amd extensions
                         mode="auto", mano_model=None, rend_size=256, ordinal_depth=False):
    auto = mode == "auto"
    if auto:
        # the fused launch sequence whenever FusedStepper accepts the configuration (every BASELINE config), else the same
        # iteration through HOMan.forward + autograd in a hipGraph; mode="eager" is the reference's loop verbatim (host sync
        # per logged value).  FusedStepper's own capability guards decide (below): no second copy of them here.
        mode = "fused"
    model = build_model(person_parameters, object_parameters, class_name, objvertices, objfaces, camintr,
                        hand_proj_mode, optimize_mano, optimize_mano_beta, optimize_object_scale, state_dict,
                        image_size, mano_model, rend_size, sync_metrics=(mode == "eager"), ordinal_depth=ordinal_depth)
    # visualisation frames every `viz_step` iterations (reference jointopt.py:158-176), only when the caller hands the
    # input images over; the videos the reference assembles from them (libyana np2vid, :193-200) are not built
    imgs = OrderedDict()
    viz = images is not None and viz_step
    if mode in ("graph", "fused"):
        if mode == "fused":
            try:
                stepper = FusedStepper(model, loss_weights, lr, num_iterations)
            except NotImplementedError:       # (the capability guards only: a library error is a bug and propagates)
                if not auto:
                    raise
                mode = "graph"        # a configuration the fused launch sequence does not cover
        if mode == "graph":
            stepper = GraphStepper(model, loss_weights, lr, num_iterations)
        step = 0
        while step < num_iterations:
            if viz:
                imgs[step] = save_front_top(model, images, step, viz_folder, viz_len)
            chunk = min(viz_step, num_iterations - step) if viz else num_iterations
            stepper.run(chunk)
            step += chunk
        return model, stepper.loss_evolution(num_iterations), imgs
    if mode != "eager":
        raise ValueError(f"mode {mode} not in [auto|eager|graph|fused]")
    optimizer = torch.optim.Adam(parameter_groups(model, lr))
    loss_evolution = defaultdict(list)
    for step in range(num_iterations):
        if viz and step % viz_step == 0:
            imgs[step] = save_front_top(model, images, step, viz_folder, viz_len)
        optimizer.zero_grad()
        loss_dict, metric_dict = model(loss_weights=loss_weights)
        loss_dict_weighted = {k: loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict}
        for k, val in loss_dict.items():
            loss_evolution[k].append(val.item())
        for k, val in metric_dict.items():
            loss_evolution[k].append(val)
        loss = sum(loss_dict_weighted.values())
        loss_evolution["loss"].append(loss.item())
        loss.backward()
        optimizer.step()
    return model, dict(loss_evolution), imgs
