"""`--hand_proj_mode ortho` (reference homan/homan.py:364-371 -> utils/camera.py:59-105): the hand placed by its scaled-orthographic
camera instead of rotation + translation.  The camera conversion is a libyana function that is not in /root/reference, so the
mode is PARITY-UNPINNED: what is tested is the camera identity it stands for (known answers), oracle == HIP, and that the
reference loop's fallback (FusedStepper refuses the mode -> autograd in a hipGraph) fits."""
import copy

import numpy as np
import pytest
import torch

from tests import util


def test_weak_camera_known_answers():
    from homan_amd.homan import weakcam_persp_trans
    from homan_amd.synth import weakcams_from_translations
    from oracle import yana
    g = torch.Generator().manual_seed(3)
    B = 5
    K = torch.tensor([[[1.2, 0.0, 0.45], [0.0, 1.1, 0.55], [0.0, 0.0, 1.0]]]).repeat(B, 1, 1)
    cams = torch.stack([0.5 + torch.rand(B, generator=g), torch.rand(B, generator=g) - 0.5, torch.rand(B, generator=g) - 0.5], 1)
    T = weakcam_persp_trans(cams, K)[:, 0]
    # the mesh origin lands on the weak camera's pixel translation under the pinhole camera, and a small offset moves
    # by the weak camera's pixel scale
    pscale = cams[:, 0] / 2 * 640
    ptrans = (cams[:, 1:] + 1 / cams[:, :1]) * pscale[:, None]
    Kp = K.clone()
    Kp[:, :2] *= 640
    proj = lambda X: torch.stack([Kp[:, 0, 0] * X[:, 0] / X[:, 2] + Kp[:, 0, 2], Kp[:, 1, 1] * X[:, 1] / X[:, 2] + Kp[:, 1, 2]], 1)
    np.testing.assert_allclose(proj(T).numpy(), ptrans.numpy(), rtol=2e-6)
    dx = torch.tensor([1e-3, 0.0, 0.0])
    np.testing.assert_allclose(((proj(T + dx) - proj(T))[:, 0] / 1e-3).numpy(), pscale.numpy(), rtol=2e-3)
    # the oracle's statement of the libyana call gives the same translation
    To = yana.batch_weakcam2persptrans(torch.cat([pscale[:, None], ptrans], 1), Kp, 1)
    np.testing.assert_allclose(To.numpy(), T.numpy(), rtol=1e-6)
    # and the synthetic-input helper inverts it
    np.testing.assert_allclose(weakcams_from_translations(T, K).numpy(), cams.numpy(), rtol=1e-5, atol=1e-6)


def _ortho_clip(mano_model, frames=4, size=64):
    from homan_amd import synth
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=11, frames=frames, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    for i, p in enumerate(clip["person_parameters"]):       # (one hand per frame: person i is frame i)
        p["cams"] = synth.weakcams_from_translations(p["translations"], clip["camintr"][i:i + 1])
    return clip


def test_oracle_ortho_mode_gradients(mano_model):
    """oracle model: the camera gets the gradient rotation / translation get in the perspective mode; the twin detaches the mesh only."""
    from homan_amd import synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan, transform_ortho
    clip = _ortho_clip(mano_model)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    om = OracleHOMan(**copy.deepcopy(kw), camintr=clip["camintr"], optimize_mano=True, image_size=64, mano_model=mano_model,
                     rend_size=64, hand_proj_mode="ortho")
    # with an identity hand rotation the ortho placement equals the perspective one it was derived from
    v_o = om.get_verts_hand()[0]
    with torch.no_grad():
        om.rotations_hand.copy_(torch.eye(3)[:, :2].expand_as(om.rotations_hand))
    om.hand_proj_mode = "persp"
    v_p = om.get_verts_hand()[0]
    om.hand_proj_mode = "ortho"
    np.testing.assert_allclose(v_o.detach().numpy(), v_p.detach().numpy(), atol=2e-6)
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    ld, _ = om(loss_weights=lw)
    sum(ld[k] * lw[k.replace("loss", "lw")] for k in ld).sum().backward()
    assert om.cams_hand.grad is not None and om.cams_hand.grad.abs().max() > 0
    assert om.rotations_hand.grad is None and om.translations_hand.grad is None
    mesh = torch.randn(2, 7, 3, requires_grad=True)
    s = torch.ones(1, requires_grad=True)
    cams = torch.tensor([[1.0, 0.1, 0.0], [0.8, 0.0, 0.2]], requires_grad=True)
    full, twin = transform_ortho(mesh, cams, s, om.camintr[:1].expand(2, 3, 3))
    assert torch.equal(full, twin)
    twin.sum().backward()
    assert mesh.grad is None and s.grad.abs().sum() > 0 and cams.grad.abs().sum() > 0


@pytest.mark.gpu
def test_hip_ortho_mode_matches_oracle_and_fits(mano_model):
    from homan_amd import HOMan, synth
    from homan_amd.jointopt import optimize_hand_object
    from oracle.jointopt import collate_inputs
    from oracle.jointopt import optimize_hand_object as oracle_fit
    from oracle.model import OracleHOMan
    clip = _ortho_clip(mano_model)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=64,
                  mano_model=mano_model, rend_size=64, hand_proj_mode="ortho")
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    om = OracleHOMan(**copy.deepcopy(kw), **common)
    hm = HOMan(**copy.deepcopy(kw), **common)
    dv = (hm.get_verts_hand()[0].detach().cpu() - om.get_verts_hand()[0].detach()).abs().max().item()
    assert dv < 1e-6, dv          # metres: 1e-3 mm
    lo, mo = om(loss_weights=lw)
    lh, mh = hm(loss_weights=lw)
    for k in lo:
        np.testing.assert_allclose(lh[k].detach().cpu().numpy(), lo[k].detach().numpy(), rtol=1e-4, atol=1e-9, err_msg=k)
    sum(lo[k] * lw[k.replace("loss", "lw")] for k in lo).sum().backward()
    sum(lh[k] * lw[k.replace("loss", "lw")] for k in lh).sum().backward()
    for name in ("cams_hand", "mano_pca_pose", "mano_rot", "mano_trans", "mano_betas", "translations_object", "rotations_object"):
        go, gh = getattr(om, name).grad, getattr(hm, name).grad
        assert go is not None and gh is not None, name
        scale = go.abs().max().item()
        np.testing.assert_allclose(gh.cpu().numpy() / scale, go.numpy() / scale, atol=2e-4, err_msg=name)
    assert hm.rotations_hand.grad is None and hm.translations_hand.grad is None
    # the loop: FusedStepper refuses the mode, mode="auto" falls back to the autograd hipGraph; the first steps follow the oracle's loop
    args = dict(class_name="default", objvertices=clip["objvertices"], objfaces=clip["objfaces"], loss_weights=lw,
                num_iterations=6, lr=1e-2, camintr=clip["camintr"], hand_proj_mode="ortho", optimize_mano=True, image_size=64,
                mano_model=mano_model, rend_size=64)
    _, evo_o = oracle_fit(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), **args)[:2]
    model, evo_h, _ = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), **args)
    np.testing.assert_allclose(evo_h["loss"][0], evo_o["loss"][0], rtol=1e-4)
    np.testing.assert_allclose(evo_h["loss"][:3], evo_o["loss"][:3], rtol=5e-3)
    assert evo_h["loss"][-1] < evo_h["loss"][0]
    assert (model.cams_hand.detach().cpu() - kw["cams_hand"]).abs().max() > 0       # the camera moved


@pytest.mark.gpu
def test_clip_fitter_takes_ortho_clips_through_the_graph_loop(mano_model):
    """A dataset walk in ortho mode: the fused loop refuses the mode for batches and for single clips, ClipFitter then fits
    every clip through the autograd hipGraph - the result of optimize_hand_object for that clip."""
    from homan_amd import synth
    from homan_amd.jointopt import ClipFitter, optimize_hand_object
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    clips = [_ortho_clip(mano_model), _ortho_clip(mano_model, frames=3)]
    fitter = ClipFitter(lw, num_iterations=5, lr=1e-2, clips_per_batch=2, hand_proj_mode="ortho", optimize_mano=True,
                        image_size=64, mano_model=mano_model, rend_size=64)
    results = fitter.fit(copy.deepcopy(clips) + [copy.deepcopy(clips[0])])
    assert len(results) == 3 and not fitter.resident
    # one resident autograd graph per shape (two shapes), the third clip reloaded into the first one's - not a graph per clip
    assert len(fitter.resident_graph) == 2 and fitter.timing["built"] == 2 and fitter.timing["reused"] == 1
    for clip, res in zip(clips + [clips[0]], results):
        model, evo, _ = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                             objvertices=clip["objvertices"], objfaces=clip["objfaces"], loss_weights=lw,
                                             num_iterations=5, lr=1e-2, camintr=clip["camintr"], hand_proj_mode="ortho",
                                             optimize_mano=True, image_size=64, mano_model=mano_model, rend_size=64)
        np.testing.assert_array_equal(res["loss_evolution"]["loss"], evo["loss"])
        assert torch.equal(res["state_dict"]["cams_hand"], model.cams_hand.detach().cpu())
        assert torch.equal(res["verts_hand"], model.get_verts_hand()[0].detach().cpu())
