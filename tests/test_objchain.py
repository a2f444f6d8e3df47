"""The written-out object chain (oracle/objchain.py, oracle/csrc/objchain.c, oracle/adam.py) against the faithful
restatement it must agree with - autograd through OracleHOMan.forward (pinned to the reference's goldens by
tests/test_oracle_golden.py) and torch.optim.Adam - and the property it exists for: its results do not depend on the
number of host threads.  CPU only."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _clip_model(mano_model, seed=0, frames=4, size=64, obj="cube", hands=("right",)):
    from homan_amd import synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn, hands=hands)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    return OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                       image_size=size, mano_model=mano_model, rend_size=size, **kw), clip


@pytest.mark.parametrize("obj", ["cube", "bottle"])
def test_written_out_chain_equals_autograd(obj, mano_model):
    from homan_amd import synth
    from oracle import objchain
    model, _ = _clip_model(mano_model, seed=3, obj=obj)
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    loss_dict, _ = model(loss_weights=lw)
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    got = objchain.object_pose_grads(model, lw)
    for name in ("rotations_object", "translations_object"):
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        assert scale > 0
        np.testing.assert_allclose(got[name].reshape(ref.shape) / scale, ref / scale, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("weights_name", ["STEP1_LOSS_WEIGHTS", "CFG1_LOSS_WEIGHTS"])
def test_written_out_hand_chain_equals_autograd(weights_name, mano_model):
    """oracle/handchain.py (2-D reprojection, smoothness, interaction, PCA prior -> rigid backward -> MANO backward, one stated
    evaluation order) against autograd through the torch restatement of the same model: every hand parameter's gradient within
    fp32 rounding of its largest entry."""
    from homan_amd import synth
    from oracle import handchain
    model, _ = _clip_model(mano_model, seed=2, obj="bottle")
    with torch.no_grad():           # off the initial pose: non-zero shape and PCA coefficients, a model-space translation
        g = torch.Generator().manual_seed(0)
        model.mano_betas.add_(0.3 * torch.randn(model.mano_betas.shape, generator=g))
        model.mano_pca_pose.add_(0.2 * torch.randn(model.mano_pca_pose.shape, generator=g))
        model.mano_trans.add_(0.01 * torch.randn(model.mano_trans.shape, generator=g))
    lw = dict(getattr(synth, weights_name))
    loss_dict, _ = model(loss_weights=lw)
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    got = handchain.hand_param_grads(model, lw)
    assert sorted(got) == ["mano_betas", "mano_pca_pose", "mano_rot", "mano_trans", "rotations_hand", "translations_hand"]
    for name, g in got.items():
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        assert scale > 0, name
        np.testing.assert_allclose(g.reshape(ref.shape) / scale, ref / scale, atol=2e-5, err_msg=name)
    with pytest.raises(NotImplementedError):
        handchain.hand_param_grads(model, dict(synth.STEP2_LOSS_WEIGHTS, lw_depth=1.0))


def test_written_out_step2_terms_equal_autograd(mano_model):
    """The step-2 terms written out (oracle/handchain.py pair_terms: nearest object vertex, contact, collision samples) against
    autograd through the faithful restatement, with the hand pushed into the object so that the collision term is live.
    The written-out search differences the coordinates before squaring (as the kernels do); the reference's |a|^2 + |b|^2 - 2ab
    (contactloss.py:60-79; its rounding error at |a|^2 ~ 0.36 m^2 is ~4e-8 m^2) names another neighbour for a few hand
    vertices with two object vertices at almost the same distance: the contact term's own gradient on the object's rotation
    differs by ~2 % of ITS largest entry there - 1e-4 of the whole gradient.  That is the tolerance below."""
    from homan_amd import synth
    from oracle import handchain, objchain
    from oracle import yana as o_yana
    model, _ = _clip_model(mano_model, seed=2, obj="bottle")
    with torch.no_grad():
        model.translations_hand.add_(torch.tensor([0.03, 0.0, 0.0]))
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    loss_dict, _ = model(loss_weights=lw)
    assert float(loss_dict["loss_collision"].detach()) > 0
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    got, stg = handchain.hand_param_grads(model, lw, return_stages=True)
    got.update(objchain.object_pose_grads(model, lw, contact_obj=stg["pair"]["con_obj"]))
    for name, g in got.items():
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(g.reshape(ref.shape) / scale, ref / scale, atol=3e-4 if "object" in name else 5e-5, err_msg=name)
    # the picks themselves: the same vertex, or one at the same distance to within the expansion's rounding
    with torch.no_grad():
        vh, vo = model.get_verts_hand()[0], model.get_verts_object()[0]
        ref_idx = torch.min(o_yana.batch_pairwise_dist(vh, vo), 2)[1].numpy()
    idx = stg["pair"]["nn_idx"]
    other = np.nonzero(idx != ref_idx)
    assert len(other[0]) < 0.05 * idx.size
    d2 = lambda ii: ((stg["vh"][other[0], other[1]].astype(np.float64) - stg["vo"][other[0], ii[other]]) ** 2).sum(-1)
    assert np.all(np.abs(d2(idx) - d2(ref_idx)) < 1e-7) and np.all(d2(idx) <= d2(ref_idx) + 1e-12)       # (squared metres)
    # tanh as a defined function: within half an ulp of libm's in double
    from oracle import clib
    xs = np.concatenate([np.linspace(0, 12, 50001), [1e-9, 25.0]]).astype(np.float32)
    err = max(abs(float(clib.lib().orc_tanh(float(x))) - np.tanh(np.float64(x))) / max(np.tanh(np.float64(x)), 1e-30) for x in xs[1:])
    assert err < 6.1e-8, err


def test_written_out_mano_layer_equals_the_torch_restatement(mano_model):
    """oracle/csrc/lbs_exact.c (the evaluation order shared with csrc/mano.hip) vs oracle/lbs.py (torch): the same vertices to
    an ulp, right and left hands; and its sin / cos against libm in double."""
    from oracle import clib
    from oracle import model as o_model
    from homan_amd import synth
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=4, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn, hand_verts_fn=hand_fn,
                           hands=("right", "left"))
    from oracle.jointopt import collate_inputs
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    model = o_model.OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=64,
                                mano_model=mano_model, rend_size=64, **kw)
    with torch.no_grad():
        model.mano_betas.add_(0.3 * torch.randn(model.mano_betas.shape, generator=torch.Generator().manual_seed(1)))
        a = model.get_verts_hand()[0].numpy().copy()
        o_model.REFERENCE_FORM = True
        try:
            b = model.get_verts_hand()[0].numpy().copy()
        finally:
            o_model.REFERENCE_FORM = False
    assert np.abs(a - b).max() < 2.5e-7 and np.abs(a - b).max() > 0          # (two evaluation orders of fp32: not the same bits)
    ang = np.concatenate([np.linspace(-7, 7, 20001), [0.0, 1e-8, 1e3, -1e4]]).astype(np.float32)
    s, c = np.empty(1, np.float32), np.empty(1, np.float32)
    err = 0.0
    for x in ang:
        clib.lib().orc_sincos(float(x), clib.fptr(s), clib.fptr(c))
        err = max(err, abs(float(s[0]) - np.sin(np.float64(x))), abs(float(c[0]) - np.cos(np.float64(x))))
    assert err < 6.1e-8, err             # half an ulp of fp32 at 1


def test_written_out_chain_with_a_free_object_scale(mano_model):
    """optimize_object_scale=True: the interaction term reaches the object (homan/homan.py:482-490), the scale gets the frames'
    partial sums + its prior (homan/lossutils.py:107-109) - all nine parameters against autograd."""
    from homan_amd import synth
    from oracle import handchain, objchain
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=2, frames=4, rend_size=64, image_size=64, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    model = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                        optimize_object_scale=True, image_size=64, mano_model=mano_model, rend_size=64, **kw)
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    with torch.no_grad():
        model.translations_hand.add_(torch.tensor([0.03, 0.0, 0.0]))
        model.int_scales_object.add_(0.05)
    loss_dict, _ = model(loss_weights=lw)
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    got, stg = handchain.hand_param_grads(model, lw, return_stages=True)
    got.update(objchain.object_pose_grads(model, lw, contact_obj=stg["pair"]["con_obj"], inter_rec=stg["rec"]))
    assert len(got) == 9
    for name, g in got.items():
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(g.reshape(ref.shape) / scale, ref / scale, atol=3e-4 if "rotations_object" in name else 5e-5,
                                   err_msg=name)


def test_written_out_chain_with_inter_type_min_and_a_free_object_scale(mano_model):
    """inter_type="min" (homan/losses.py:219-221) with optimize_object_scale=True: the closest pair's pull reaches the OBJECT'S
    vertex j* too, and through it pose and scale - the last configuration the written-out chain used to refuse.  All nine
    parameters against autograd through the faithful form."""
    from homan_amd import synth
    from oracle import handchain, objchain
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=2, frames=4, rend_size=64, image_size=64, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    model = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                        optimize_object_scale=True, inter_type="min", image_size=64, mano_model=mano_model, rend_size=64, **kw)
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    with torch.no_grad():
        model.int_scales_object.add_(0.05)
    loss_dict, _ = model(loss_weights=lw)
    assert float(loss_dict["loss_inter"].detach()) > 0                     # (some frame passes the gate: the term is live)
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    got, stg = handchain.hand_param_grads(model, lw, return_stages=True)
    got.update(objchain.object_pose_grads(model, lw, inter_rec=stg["rec"]))
    assert len(got) == 9
    for name, g in got.items():
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(g.reshape(ref.shape) / scale, ref / scale, atol=3e-4 if "rotations_object" in name else 5e-5,
                                   err_msg=name)


@pytest.mark.parametrize("free_scale", [False, True])
def test_written_out_chain_with_two_hands(free_scale, mano_model):
    """hand_nb = 2 (right + left), step-2 set, both hands moved into the object: the written-out chains (oracle/handchain.py
    two_hand_terms: per-hand search / contact / interaction, three collision scenes; the hands' rigid backward; the MANO
    backward per hand through its side's model) against autograd through the faithful restatement."""
    from homan_amd import synth
    from oracle import handchain, objchain
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=4, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn, hand_verts_fn=hand_fn,
                           hands=("right", "left"))
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    model = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                        optimize_object_scale=free_scale, image_size=64, mano_model=mano_model, rend_size=64, **kw)
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    with torch.no_grad():
        model.mano_betas.add_(0.2 * torch.randn(model.mano_betas.shape, generator=torch.Generator().manual_seed(0)))
        model.translations_hand.add_(0.5 * (model.translations_object.repeat_interleave(2, 0) - model.translations_hand))
    loss_dict, _ = model(loss_weights=lw)
    assert float(loss_dict["loss_collision"].detach()) > 0
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).backward()
    two = handchain.two_hand_terms(model, lw)
    got = handchain.hand_param_grads(model, lw, two=two)
    got.update(objchain.object_pose_grads(model, lw, obj_terms=two["obj_terms"]))
    assert len(got) == (9 if free_scale else 8)
    for name, g in got.items():
        ref = getattr(model, name).grad.numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(g.reshape(ref.shape) / scale, ref / scale, atol=3e-4 if "rotations_object" in name else 1e-4,
                                   err_msg=name)


@pytest.mark.parametrize("weights_name", ["depth only", "STEP2_LOSS_WEIGHTS"])
def test_written_out_chain_with_two_hands_and_the_depth_term(weights_name, mano_model):
    """hand_nb = 2 with ordinal_depth=True (three layers, three pairs, one normaliser: reference homan/homan.py:384-419,
    lossutils.py:133-169): oracle/depthchain.py depth_vertex_grads_layers folded into the two-hand chains, against autograd
    through the faithful restatement - the depth term alone, then next to the step-2 set (the collision terms and the depth
    term reach the hands through one rigid backward)."""
    from homan_amd import synth
    from oracle.jointopt import collate_inputs, reproducible_grads
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=4, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn, hand_verts_fn=hand_fn,
                           hands=("right", "left"))
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):      # masks that disagree with the geometry
        op["full_mask"] = ((pp["masks"].sum(0) > 0) | (op["full_mask"] > 0)).float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([[[0.05, 0.0, -0.02]], [[-0.05, 0.0, -0.02]]])
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=64,
                  mano_model=mano_model, rend_size=64, ordinal_depth=True)
    lw = (dict({k: 0.0 for k in synth.STEP1_LOSS_WEIGHTS}, lw_depth=1.0) if weights_name == "depth only" else
          dict(getattr(synth, weights_name), lw_depth=2.0))
    model = OracleHOMan(**copy.deepcopy(kw), **common)
    loss_dict, _ = model(loss_weights=lw)
    assert float(loss_dict["loss_depth"].detach()) > 0
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).sum().backward()
    written = OracleHOMan(**copy.deepcopy(kw), **common)
    reproducible_grads(written, lw)
    names = [k for k, p in written.named_parameters() if p.grad is not None]
    assert len(names) == 8
    for name in names:
        ref, g = getattr(model, name).grad.numpy(), getattr(written, name).grad.numpy()
        scale = np.abs(ref).max()
        assert scale > 0, name
        np.testing.assert_allclose(g / scale, ref / scale, atol=3e-4 if "rotations_object" in name else 1e-4, err_msg=name)


def test_exact_pseudo_gradient_equals_the_faithful_loop(mano_model):
    """per (face, corner): the exact-sum variant against orc_nmr_grad_faces_alpha (the published loop order, fp32 sums)"""
    from homan_amd import synth
    from oracle import clib, objchain
    model, _ = _clip_model(mano_model, seed=5, obj="bottle")
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    _, st = objchain.object_pose_grads(model, lw, return_stages=True)
    f, idx, ga = st["faces_ndc"], st["idx"], st["grad_alpha"]
    B, NF = f.shape[:2]
    gf = np.zeros((B, NF, 9), np.float32)
    clib.lib().orc_nmr_grad_faces_alpha(clib.fptr(f), clib.iptr(idx), clib.fptr(ga), B, NF, idx.shape[1], 1e-3, clib.fptr(gf))
    gf = gf.reshape(B, NF, 3, 3)[..., :2]
    F = NF // 2
    ref = gf[:, :F] + gf[:, F:, ::-1]                 # corner k of the reversed copy is mesh corner 2 - k
    scale = np.abs(ref).max()
    assert scale > 0
    np.testing.assert_allclose(st["parts"] / scale, ref / scale, atol=2e-6)


def test_written_out_adam_equals_torch_adam():
    from oracle.adam import Adam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ta = torch.optim.Adam([{"params": a[:1], "lr": 1e-2}, {"params": a[1:], "lr": 1e-1}])
    tb = Adam([{"params": b[:1], "lr": 1e-2}, {"params": b[1:], "lr": 1e-1}])
    for step in range(25):
        gs = [torch.randn_like(p) * (10.0 ** (step % 5 - 3)) for p in a]
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        ta.step()
        tb.step()
        for p, q in zip(a, b):
            np.testing.assert_allclose(q.detach().numpy(), p.detach().numpy(), rtol=2e-6, atol=1e-7)


_THREADS_SCRIPT = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from homan_amd.mano_assets import synthetic_mano
from homan_amd import synth
from tests.test_objchain import _clip_model
from oracle.jointopt import make_optimizer, reproducible_step
mano = synthetic_mano(0)
torch.set_num_threads(1)            # (the INPUTS - 2-D targets projected with torch - are made the same way in both runs)
model, _ = _clip_model(mano, seed=1, frames=4, size=64, obj="cube", **(dict(hands=("right", "left")) if sys.argv[3].endswith("+2") else dict()))
sys.argv[3] = sys.argv[3].replace("+2", "")
torch.set_num_threads(int(sys.argv[1]))
lw = dict(getattr(synth, sys.argv[3]))
opt = make_optimizer(model, 1e-2, reproducible=True)
for _ in range(12):
    reproducible_step(model, lw, opt)
np.save(sys.argv[2], np.concatenate([p.detach().numpy().ravel() for _, p in sorted(model.named_parameters())]))
"""


@pytest.mark.parametrize("weights_name", ["CFG1_LOSS_WEIGHTS", "STEP1_LOSS_WEIGHTS", "STEP2_LOSS_WEIGHTS", "STEP2_LOSS_WEIGHTS+2"])
def test_trajectory_does_not_depend_on_the_thread_count(weights_name, tmp_path):
    """the oracle's end state was a function of OMP_NUM_THREADS.  With the written-out chains (object: oracle/
    objchain.py, hand: oracle/handchain.py) EVERY parameter after 12 steps is bit-identical at 1 and 4 threads."""
    outs = []
    for nt in (1, 4):
        out = str(tmp_path / f"p{nt}.npy")
        env = dict(os.environ, OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt))
        subprocess.run([sys.executable, "-c", _THREADS_SCRIPT.format(root=ROOT), str(nt), out, weights_name], check=True, env=env,
                       cwd=ROOT)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])
