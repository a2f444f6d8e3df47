"""Generate tests/golden/*.npz by running the REFERENCE's own composition code
(/root/reference/homan/{homan,losses,lossutils,jointopt,...}.py, imported in place) over the
repo's CPU oracle leaves.  Build-container only (needs /root/reference); the vectors are data:
inputs + the reference's outputs.  Usage:  python tools/refharness/gen_goldens.py
"""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import shims  # noqa: E402

ref_homan, ref_jointopt, ref_lossutils = shims.install()
from oracle import lbs, nmr  # noqa: E402
from oracle.jointopt import collate_inputs  # noqa: E402
from homan_amd import synth  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

OUT = os.path.join(shims.REPO_ROOT, "tests", "golden")
MANO = synthetic_mano(0)
from homan_amd.mano_assets import hand_models  # noqa: E402
_layers = {side: lbs.ManoLayer(m, num_pca_comps=16, flat_hand_mean=False) for side, m in hand_models(MANO).items()}


def hand_fn(pca, rot, betas, side="right"):
    hp = pca[:, :16] @ _layers[side].hand_components
    if side == "left":      # the ground-truth pose only: the reference's own left path runs inside its HOMan below
        hp = hp.clone()
        hp[:, 1::3] *= -1
        hp[:, 2::3] *= -1
    return _layers[side](betas=betas, global_orient=rot, hand_pose=hp, transl=torch.zeros(len(rot), 3))[0]


def sil_fn(verts, faces, K, size):
    r = nmr.Renderer(image_size=size, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    return r(verts, faces, mode="silhouettes")


def flat_inputs(clip):
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"],
                        clip["objfaces"])
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            out["in_" + k] = v.numpy()
    out["in_camintr"] = np.asarray(clip["camintr"], np.float32)
    return out


def run_case(name, seed, frames, size, obj, weights, steps, optimize_object_scale=False,
             optimize_mano=True, init_steps=0, pin_step=5, hands=("right",), inter_type="centroid", fwd_only=False):
    shims.set_rend_size(size)
    clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj,
                           silhouette_fn=sil_fn, hand_verts_fn=hand_fn, hands=hands)
    rec = flat_inputs(clip)
    rec["meta_image_size"] = np.int64(size)
    rec["meta_steps"] = np.int64(steps)
    rec["meta_optimize_object_scale"] = np.int64(optimize_object_scale)
    rec["meta_optimize_mano"] = np.int64(optimize_mano)
    rec["meta_lr"] = np.float64(1e-2)
    rec["meta_hand_sides"] = np.array(list(hands))
    rec["meta_inter_type"] = np.array(inter_type)
    for k, v in weights.items():
        rec["lw_" + k[3:]] = np.float64(v)

    common = dict(objvertices=clip["objvertices"], objfaces=clip["objfaces"], lr=1e-2,
                  camintr=clip["camintr"], optimize_mano=optimize_mano,
                  optimize_object_scale=optimize_object_scale, image_size=size, viz_step=10 ** 6)
    # single forward/backward at the initial state: losses, metrics, grads of every Parameter
    model, _, _ = ref_jointopt.optimize_hand_object(
        copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
        loss_weights=weights, num_iterations=1, viz_folder=tempfile.mkdtemp(), **common)
    # (the loop above already took one Adam step; rebuild an un-stepped model for the gradient record)
    kw = collate_inputs(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        clip["objvertices"], clip["objfaces"])
    fresh = ref_homan.HOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1,
                            hand_proj_mode="persp", optimize_mano=optimize_mano, optimize_mano_beta=True,
                            optimize_object_scale=optimize_object_scale, image_size=size, inter_type=inter_type, **kw)
    loss_dict, metric_dict = fresh(loss_weights=weights)
    total = sum(loss_dict[k] * weights[k.replace("loss", "lw")] for k in loss_dict)
    total.backward()
    for k, v in loss_dict.items():
        rec["fwd_" + k] = v.detach().numpy()
    for k, v in metric_dict.items():
        rec["metric_" + k] = np.float64(v)
    for k, p in fresh.named_parameters():
        rec["grad_" + k] = (p.grad.numpy() if p.grad is not None else np.zeros(0, np.float32))
    vo, _ = fresh.get_verts_object()
    vh, _ = fresh.get_verts_hand()
    rec["verts_object"] = vo.detach().numpy()
    rec["verts_hand"] = vh.detach().numpy()
    rec["state_dict_keys"] = np.array(sorted(fresh.state_dict().keys()))
    if inter_type != "centroid" or fwd_only:
        # the reference's loop (jointopt.optimize_hand_object) builds its HOMan with the default interaction term: a non-default
        # `inter_type` is pinned by the forward / backward of the model alone.  `fwd_only`: the BASELINE-sized clips
        # (30 frames 256^2), where one forward / backward of the reference over the CPU leaves is what a test can afford
        out_dir = os.environ.get("HOMAN_GOLDEN_OUT", OUT)
        os.makedirs(out_dir, exist_ok=True)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "fwd only, loss_inter", rec["fwd_loss_inter"])
        return

    # trajectory with the reference's own loop
    model, evo, _ = ref_jointopt.optimize_hand_object(
        copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
        loss_weights=weights, num_iterations=steps, viz_folder=tempfile.mkdtemp(), **common)
    for k, v in evo.items():
        rec["evo_" + k] = np.asarray(v, np.float64)
    for k, v in model.state_dict().items():
        if k in dict(model.named_parameters()) or k.startswith("int_scales"):
            rec["final_" + k] = v.detach().numpy()
    # per-step pin: the reference loop's parameters after `pin_step` steps, and ONE forward / backward of the
    # reference model AT those parameters.  A restatement loaded with the same parameters must reproduce these at single-step
    # tolerance - a semantic error that only shows once the optimiser has moved (Adam state aside) cannot hide behind the
    # chaotic separation of long trajectories.
    pin_step = min(pin_step, steps)
    model_k, _, _ = ref_jointopt.optimize_hand_object(
        copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
        loss_weights=weights, num_iterations=pin_step, viz_folder=tempfile.mkdtemp(), **common)
    pinned = {k: v.detach().clone() for k, v in model_k.named_parameters()}
    at_k = ref_homan.HOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1,
                           hand_proj_mode="persp", optimize_mano=optimize_mano, optimize_mano_beta=True,
                           optimize_object_scale=optimize_object_scale, image_size=size,
                           **collate_inputs(copy.deepcopy(clip["person_parameters"]),
                                            copy.deepcopy(clip["object_parameters"]), clip["objvertices"],
                                            clip["objfaces"]))
    at_k.load_state_dict(pinned, strict=False)
    ld, md = at_k(loss_weights=weights)
    sum(ld[k] * weights[k.replace("loss", "lw")] for k in ld).backward()
    rec["meta_pin_step"] = np.int64(pin_step)
    for k, v in pinned.items():
        rec["pin_" + k] = v.numpy()
    for k, v in ld.items():
        rec["pinfwd_" + k] = v.detach().numpy()
    for k, v in md.items():
        rec["pinmetric_" + k] = np.float64(v)
    for k, p in at_k.named_parameters():
        rec["pingrad_" + k] = (p.grad.numpy() if p.grad is not None else np.zeros(0, np.float32))
    out_dir = os.environ.get("HOMAN_GOLDEN_OUT", OUT)
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(name, "loss", evo["loss"][0], "->", evo["loss"][-1])


def _maybe(name, **kw):
    if len(sys.argv) < 2 or sys.argv[1] in name:
        run_case(name, **kw)


def main():
    _maybe("ref_step1_cube_b4_s64", seed=0, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP1_LOSS_WEIGHTS), steps=20)
    _maybe("ref_step2_cube_b4_s64", seed=1, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP2_LOSS_WEIGHTS), steps=10)
    # NB: never use frames == 3: the reference's dim-less torch.cross (homan/utils/geometry.py:26)
    # silently takes the cross product along the batch axis when the flattened batch is exactly 3.
    _maybe("ref_step2_scale_bottle_b5_s64", seed=2, frames=5, size=64, obj="bottle",
             weights=dict(synth.STEP2_LOSS_WEIGHTS), steps=6, optimize_object_scale=True)
    _maybe("ref_rigid_cube_b5_s32", seed=3, frames=5, size=32, obj="cube",
             weights=dict(synth.STEP1_LOSS_WEIGHTS), steps=10, optimize_mano=False)
    # two hands, right + left (hand_nb = 2): every loss of the refinement step, so that the multi-hand branches of
    # lossutils.py:51-64,114-131 (collision over three meshes, contact per hand) and the left MANO path are pinned
    _maybe("ref_step2_twohands_cube_b4_s64", seed=4, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP2_LOSS_WEIGHTS), steps=8, hands=("right", "left"))
    # non-default interaction term of the model (homan/losses.py:219-221): smallest squared vertex distance
    _maybe("ref_step2_intermin_cube_b4_s64", seed=1, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP2_LOSS_WEIGHTS), steps=1, inter_type="min")
    _maybe("ref_step1_lefthand_cube_b4_s64", seed=6, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP1_LOSS_WEIGHTS), steps=8, hands=("left",))
    _maybe("ref_step1_twohands_cube_b4_s64", seed=5, frames=4, size=64, obj="cube",
             weights=dict(synth.STEP1_LOSS_WEIGHTS), steps=8, hands=("right", "left"))
    # BASELINE.json's own configurations at FULL size (the oracle <-> reference tie had only been made at 4-5
    # frames x 32-64^2).  cfg1 = configs[0], the configuration the reference's CPU path is defined on: the whole 100-step
    # fit of the reference's loop; cfg2 / cfg3 = configs[1] / [2]: one forward / backward of the reference's HOMan
    _maybe("ref_cfg1_cube_b10_s128", seed=0, frames=10, size=128, obj="cube",
             weights=dict(synth.CFG1_LOSS_WEIGHTS), steps=100)
    _maybe("ref_cfg2_bottle_b30_s256", seed=0, frames=30, size=256, obj="bottle",
             weights=dict(synth.STEP1_LOSS_WEIGHTS), steps=1, fwd_only=True)
    _maybe("ref_cfg3_bottle_b30_s256", seed=0, frames=30, size=256, obj="bottle",
             weights=dict(synth.STEP2_LOSS_WEIGHTS), steps=1, fwd_only=True)


if __name__ == "__main__":
    main()
