"""CPU restatement of the reference's optimisation loop.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
Follows reference homan/jointopt.py:22-201 minus visualisation / video export
(:159-177,193-200): per-frame dict concatenation (:55-91), HOMan construction with the
integer int_scale_init=1 (:92-124), optional state_dict resume (:126-127), three Adam
groups selected by parameter-name substring with lr, 10*lr, 10*lr (:128-151), and the
step: zero_grad, forward, loss_k * lw[k.replace('loss','lw')], per-key .item() logging,
sum, backward, step (:178-192).
"""
from collections import defaultdict

import numpy as np
import torch

from . import yana
from .model import OracleHOMan


def parameter_groups(model, lr):
    """reference homan/jointopt.py:128-151."""
    rigid = [v for k, v in model.named_parameters() if "mano" not in k and "rotation" not in k]
    rotation = [v for k, v in model.named_parameters() if ("rotation" in k) and ("mano" not in k)]
    return [{"params": rigid, "lr": lr},
            {"params": [model.mano_pca_pose, model.mano_betas], "lr": lr * 10},
            {"params": rotation, "lr": lr * 10}]


def reproducible_grads(model, loss_weights, log2q=0):
    """The gradients of one iteration of the loop below with the parts fp32 leaves open pinned down (oracle/objchain.py, oracle/adam.py):
    forward + autograd as always, then the object's pose gradients REPLACED by the written-out chain with order-independent
    sums (same mathematics, a defined rounding) and - for the loss sets oracle/handchain.py covers - the hand's by ITS written-out
    chain; they are left in the parameters' `.grad`.  With the written-out Adam (`reproducible_step`) the trajectory is then a
    function of the inputs alone - the same for any number of host threads - and bit-equal to the HIP loop's.
    -> (loss_dict, metric_dict, total)."""
    from . import objchain
    for p in model.parameters():
        p.grad = None
    obj = [model.rotations_object, model.translations_object]
    if model.optimize_object_scale:
        obj.append(model.int_scales_object)
    for p in obj:               # (their gradients come from the written-out chain below: autograd need not walk the renderer)
        p.requires_grad_(False)
    try:
        loss_dict, metric_dict = model(loss_weights=loss_weights)
        loss = sum(loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict)
        loss.backward()
    finally:
        for p in obj:
            p.requires_grad_(True)
    from . import handchain
    if model.hand_nb == 2 and (not loss_weights.get("lw_depth", 0) > 0 or getattr(model, "ordinal_depth", False)):
        # two hands per frame: the terms between the meshes per hand, then the object's and the hands' chains
        two = handchain.two_hand_terms(model, loss_weights)
        grads = objchain.object_pose_grads(model, loss_weights, log2q, obj_terms=two["obj_terms"])
        try:
            grads.update(handchain.hand_param_grads(model, loss_weights, two=two))
        except NotImplementedError:
            pass
        for k, g in grads.items():
            p = getattr(model, k)
            p.grad = torch.from_numpy(g).reshape(p.shape)
        return loss_dict, metric_dict, loss
    pair = None
    if loss_weights.get("lw_contact", 0) > 0 or loss_weights.get("lw_collision", 0) > 0:     # step-2 terms between the meshes
        with torch.no_grad():
            vh = np.ascontiguousarray(model.get_verts_hand()[0].numpy(), np.float32)
            vo = np.ascontiguousarray(model.get_verts_object()[0].numpy(), np.float32)
        pair = handchain.pair_terms(model, vh, vo, loss_weights)
    rec = None
    if model.optimize_object_scale and loss_weights.get("lw_inter", 0) > 0:     # (the term then reaches the object too)
        with torch.no_grad():
            rec = handchain.inter_records(np.ascontiguousarray(model.get_verts_hand()[0].numpy(), np.float32),
                                          np.ascontiguousarray(model.get_verts_object()[0].numpy(), np.float32),
                                          np.ascontiguousarray(model.camintr.numpy(), np.float32))
    dep_o = dep_h = None
    if loss_weights.get("lw_depth", 0) > 0 and getattr(model, "ordinal_depth", False) and model.hand_nb == 1:
        from . import depthchain
        dep_o, dep_h = depthchain.depth_vertex_grads(model, loss_weights["lw_depth"])
    grads = objchain.object_pose_grads(model, loss_weights, log2q, contact_obj=pair.get("con_obj") if pair else None,
                                       inter_rec=rec, depth_obj=dep_o)
    try:        # the hand's chain in its written-out order too, where it covers the configuration (one hand, fixed scale)
        grads.update(handchain.hand_param_grads(model, loss_weights, pair=pair, depth_hand=dep_h))
    except NotImplementedError:
        pass    # (autograd's gradients stay: same mathematics, rounding left to torch)
    for k, g in grads.items():
        p = getattr(model, k)
        p.grad = torch.from_numpy(g).reshape(p.shape)
    return loss_dict, metric_dict, loss


def reproducible_step(model, loss_weights, optimizer, log2q=0):
    """`reproducible_grads` + the written-out Adam (`optimizer` = oracle.adam.Adam).  -> (loss_dict, metric_dict, total)"""
    out = reproducible_grads(model, loss_weights, log2q)
    optimizer.step()
    return out


def reproducible_step_shared_scale(models, optimizers, loss_weights, log2q=0):
    """One step of BASELINE cfg5's loop on ONE rank: clips with ONE object scale between them (models built with
    optimize_object_scale=True, their scalars equal on entry; reference homan/jointopt.py:158-192 per clip, the tie as in
    homan_amd.dist.optimize_clips_shared_scale).  Every clip's gradients by `reproducible_grads`, the clips' d loss / d scale
    added by one 64-thread block sum (csrc/geometry.hip k_sum_small, as the fused loop adds them), the sum written to every
    replica, every clip's written-out Adam stepped - the replicas stay identical.  -> [(loss_dict, metric_dict, total)]"""
    from . import clib
    outs = [reproducible_grads(m, loss_weights, log2q) for m in models]
    g = np.ascontiguousarray([float(m.int_scales_object.grad.reshape(-1)[0]) for m in models], np.float32)
    tied = np.float32(1.0) * np.float32(clib.lib().orc_block_sum(clib.fptr(g), len(models), 64)) + np.float32(0.0)
    for m, opt in zip(models, optimizers):
        m.int_scales_object.grad = torch.full_like(m.int_scales_object, float(tied))
        opt.step()
    return outs




def collate_inputs(person_parameters, object_parameters, objvertices, objfaces):
    """reference homan/jointopt.py:52-91."""
    cat = torch.cat
    pp, op = person_parameters, object_parameters
    return dict(
        hand_sides=pp[0]["hand_side"],
        translations_object=cat([o["translations"] for o in op]),
        rotations_object=cat([o["rotations"] for o in op]),
        verts_object_og=yana.tensorify(objvertices),
        faces_object=yana.tensorify(objfaces),
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


def make_optimizer(model, lr, reproducible=False):
    """reference homan/jointopt.py:128-151.  reproducible: the written-out Adam of oracle/adam.py (see reproducible_step)."""
    if reproducible:
        from .adam import Adam
        return Adam(parameter_groups(model, lr))
    return torch.optim.Adam(parameter_groups(model, lr))


def optimize_hand_object(person_parameters, object_parameters, class_name="default", objvertices=None,
                         objfaces=None, loss_weights=None, num_iterations=400, lr=1e-2, camintr=None,
                         hand_proj_mode="persp", optimize_mano=False, optimize_mano_beta=True,
                         optimize_object_scale=False, state_dict=None, image_size=640, mano_model=None,
                         rend_size=256, log=True, ordinal_depth=False, reproducible=False):
    kw = collate_inputs(person_parameters, object_parameters, objvertices, objfaces)
    model = OracleHOMan(camintr=camintr, class_name=class_name, int_scale_init=1,
                        hand_proj_mode=hand_proj_mode, optimize_mano=optimize_mano,
                        optimize_mano_beta=optimize_mano_beta, optimize_object_scale=optimize_object_scale,
                        image_size=image_size, mano_model=mano_model, rend_size=rend_size,
                        ordinal_depth=ordinal_depth, **kw)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=False)
    optimizer = make_optimizer(model, lr, reproducible)
    loss_evolution = defaultdict(list)
    for _ in range(num_iterations):
        if reproducible:        # (same loop; the object's gradient chain and Adam in their written-out forms)
            loss_dict, metric_dict, loss = reproducible_step(model, loss_weights, optimizer)
            if log:
                for k, val in loss_dict.items():
                    loss_evolution[k].append(val.item())
                for k, val in metric_dict.items():
                    loss_evolution[k].append(val)
                loss_evolution["loss"].append(loss.item())
            continue
        optimizer.zero_grad()
        loss_dict, metric_dict = model(loss_weights=loss_weights)
        weighted = {k: loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict}
        if log:
            for k, val in loss_dict.items():
                loss_evolution[k].append(val.item())
            for k, val in metric_dict.items():
                loss_evolution[k].append(val)
        loss = sum(weighted.values())
        if log:
            loss_evolution["loss"].append(loss.item())
        loss.backward()
        optimizer.step()
    return model, dict(loss_evolution), {}
