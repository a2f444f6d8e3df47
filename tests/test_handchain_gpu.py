"""The hand's gradient chain, HIP fused step vs the CPU oracle's written-out chain (oracle/handchain.py): BIT-EQUAL gradients
for every hand parameter at identical parameters, stage by stage (unit gradients of the 2-D and smoothness terms, the
interaction term's per-frame records, then the six parameter gradients), and from there every parameter of a free-running fit.
Reference: homan/homan.py:341-382, 421-508; homan/manomodel.py:84-151; loop homan/jointopt.py:158-192.  GPU box."""
import copy

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
HAND = ("mano_pca_pose", "mano_rot", "mano_betas", "mano_trans", "rotations_hand", "translations_hand")


def _pair(mano_model, seed, frames, size, obj, hands=("right",), **options):
    from homan_amd import HOMan, synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn, hands=hands)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=size,
                  mano_model=mano_model, rend_size=size, **options)
    return HOMan(**copy.deepcopy(kw), **common), OracleHOMan(**copy.deepcopy(kw), **common)


@pytest.mark.parametrize("weights_name,obj", [("STEP1_LOSS_WEIGHTS", "bottle"), ("CFG1_LOSS_WEIGHTS", "cube"),
                                              ("STEP2_LOSS_WEIGHTS", "bottle")])
def test_hand_gradients_bit_equal(weights_name, obj, mano_model):
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    hm, om = _pair(mano_model, seed=11, frames=6, size=128, obj=obj)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():        # off the initial pose: non-zero shape / PCA coefficients / model-space translation
        moves = [("mano_betas", 0.3), ("mano_pca_pose", 0.2), ("mano_trans", 0.01), ("mano_rot", 0.05)]
        for name, amp in moves:
            d = amp * torch.randn(getattr(om, name).shape, generator=g)
            getattr(om, name).add_(d)
            getattr(hm, name).add_(d.to(getattr(hm, name).device))
        if weights_name == "STEP2_LOSS_WEIGHTS":     # the hand pushed INTO the object: the collision term is active
            d = 0.6 * (om.translations_object - om.translations_hand)
            om.translations_hand.add_(d)
            hm.translations_hand.add_(d.to(hm.translations_hand.device))
    lw = dict(getattr(synth, weights_name))
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    want, stg = handchain.hand_param_grads(om, lw, return_stages=True)
    if weights_name == "STEP2_LOSS_WEIGHTS":
        pair = stg["pair"]
        assert np.array_equal(st.nn_idx.cpu().numpy(), pair["nn_idx"])
        assert np.array_equal(st.U_conh.cpu().numpy(), pair["con_hand"]) and np.array_equal(st.U_cono.cpu().numpy(), pair["con_obj"])
        assert np.abs(pair["col_hand"]).max() > 0                      # (the term is live in this scene)
        assert np.array_equal(st.U_colh.cpu().numpy(), pair["col_hand"])
        want_o = objchain.object_pose_grads(om, lw, contact_obj=pair["con_obj"])
        for k, v in want_o.items():
            assert np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v), k
    assert np.array_equal(st.vh.cpu().numpy(), stg["vh"]) and np.array_equal(st.vm.cpu().numpy(), stg["mesh"])
    assert np.array_equal(st.vo.cpu().numpy(), stg["vo"])
    names = [n for n, on in (("U_smh", lw["lw_smooth_hand"] > 0 or lw["lw_smooth_obj"] > 0), ("U_v2d", lw["lw_v2d_hand"] > 0),
                             ("U_colh", lw["lw_collision"] > 0), ("U_conh", lw["lw_contact"] > 0)) if on]
    for n, (arr, _) in zip(names, stg["terms"]):
        assert np.array_equal(getattr(st, n).cpu().numpy(), arr), n
    if stg["rec"] is not None:
        assert np.array_equal(st.rec.cpu().numpy()[:, [0, 2, 3, 4]], stg["rec"][:, [0, 2, 3, 4]])
    report = {}
    for k in HAND:
        got = getattr(st.model, k).grad.cpu().numpy().reshape(want[k].shape)
        report[k] = (bool(np.array_equal(got, want[k])), float(np.abs(got - want[k]).max() / max(np.abs(want[k]).max(), 1e-30)))
    assert all(eq for eq, _ in report.values()), report


@pytest.mark.parametrize("weights_name", ["STEP1_LOSS_WEIGHTS", "STEP2_LOSS_WEIGHTS"])
def test_every_parameter_bit_equal_in_a_free_run(weights_name, mano_model):
    """10 free-running steps of the step-1 / step-2 loss sets (reference loop homan/jointopt.py:158-192): HIP fused loop vs the
    oracle's reproducible loop (written-out object chain, hand chain, pair terms and Adam) - EVERY parameter bit-equal after
    every step."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle.jointopt import make_optimizer, reproducible_step
    hm, om = _pair(mano_model, seed=12, frames=6, size=128, obj="bottle")
    lw = dict(getattr(synth, weights_name))
    st = FusedStepper(hm, lw, 1e-2, 6)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(6):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)


def test_free_object_scale_bit_equal(mano_model):
    """optimize_object_scale=True (BASELINE cfg5's option, one clip: the scale free): the step-2 set, gradients of all nine
    parameters - the scale's among them: the frames' exact partial sums, one block sum, the prior - bit-equal at perturbed
    parameters, then 8 free-running steps bit-equal in every parameter."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    hm, om = _pair(mano_model, seed=13, frames=6, size=128, obj="bottle", optimize_object_scale=True)
    with torch.no_grad():
        d = 0.6 * (om.translations_object - om.translations_hand)
        for m in (om, hm):
            m.translations_hand.add_(d.to(m.translations_hand.device))
            m.int_scales_object.add_(0.07)
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    want, stg = handchain.hand_param_grads(om, lw, return_stages=True)
    want.update(objchain.object_pose_grads(om, lw, contact_obj=stg["pair"]["con_obj"], inter_rec=stg["rec"]))
    assert "int_scales_object" in want
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = _pair(mano_model, seed=14, frames=6, size=128, obj="bottle", optimize_object_scale=True)
    st = FusedStepper(hm, lw, 1e-2, 12)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(8):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)
    assert abs(float(om.int_scales_object.detach()[0]) - 1.0) > 5e-4          # (the scale did move)


def _depth_pair(mano_model, seed, frames, size):
    """a clip whose instance masks DISAGREE with the initial geometry (the object annotated in front everywhere, the hand moved
    over it): the ordinal depth term is live"""
    from homan_amd import HOMan, synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
        op["full_mask"] = ((pp["masks"][0] > 0) | (op["full_mask"] > 0)).float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([0.06, 0.0, -0.02])
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=size,
                  mano_model=mano_model, rend_size=size, ordinal_depth=True)
    return HOMan(**copy.deepcopy(kw), **common), OracleHOMan(**copy.deepcopy(kw), **common)


def test_ordinal_depth_term_bit_equal(mano_model):
    """cfg2 as BASELINE.json words it (sil / kp / DEPTH / smooth): the depth term's chain - pooled depth images, per-pixel
    gradient (shared logistic function), depth-map backward per face, vertex gather - stage by stage, all eight parameter
    gradients, then 8 free-running steps bit-equal in every parameter."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import depthchain, handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(synth.STEP1_LOSS_WEIGHTS, lw_depth=1.0)
    hm, om = _depth_pair(mano_model, seed=15, frames=6, size=128)
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    dep_o, dep_h, stg = depthchain.depth_vertex_grads(om, lw["lw_depth"], return_stages=True)
    assert stg["rec"][1] + stg["rec"][3] > 0                                  # (wrongly ordered pixels exist: the term is live)
    assert np.array_equal(st.d_dep_o.cpu().numpy(), stg["pooled"][0]) and np.array_equal(st.d_dep_h.cpu().numpy(), stg["pooled"][1])
    assert np.array_equal(st.d_go.cpu().numpy(), stg["g"][0]) and np.array_equal(st.d_gh.cpu().numpy(), stg["g"][1])
    assert np.abs(dep_o).max() > 0 and np.abs(dep_h).max() > 0
    assert np.array_equal(st.G_dep_o.cpu().numpy(), dep_o) and np.array_equal(st.G_dep_h.cpu().numpy(), dep_h)
    want = handchain.hand_param_grads(om, lw, depth_hand=dep_h)
    want.update(objchain.object_pose_grads(om, lw, depth_obj=dep_o))
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = _depth_pair(mano_model, seed=16, frames=6, size=128)
    st = FusedStepper(hm, lw, 1e-2, 12)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(8):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)


def test_tied_object_scale_over_three_clips_bit_equal(mano_model):
    """BASELINE cfg5 on one rank: three clips with ONE object scale between them, step-2 loss set.  The fused loop (one clip batch,
    shared_scale=True: the clips' scale gradients added by one block sum, the sum spread to every replica) vs the oracle's
    reproducible tied loop (oracle.jointopt.reproducible_step_shared_scale): every parameter of every clip bit-equal after each
    of 6 free-running steps, the replicas of the scalar identical throughout."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle.jointopt import make_optimizer, reproducible_step_shared_scale
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    pairs = [_pair(mano_model, seed=s, frames=6, size=128, obj="bottle", optimize_object_scale=True) for s in (21, 22, 23)]
    hms, oms = [p[0] for p in pairs], [p[1] for p in pairs]
    st = FusedStepper(hms, lw, 1e-2, 10, shared_scale=True)
    opts = [make_optimizer(m, 1e-2, reproducible=True) for m in oms]
    names = [k for k, _ in oms[0].named_parameters()]
    for i in range(6):
        st.run(1)
        reproducible_step_shared_scale(oms, opts, lw)
        torch.cuda.synchronize()
        s_h = st.model.int_scales_object.detach().cpu().numpy().reshape(-1)
        assert np.all(s_h == s_h[0]) and len({float(m.int_scales_object.detach()[0]) for m in oms}) == 1
        for c, om in enumerate(oms):
            cpu = dict(om.named_parameters())
            for k in names:
                got = st.model.clip_slice(getattr(st.model, k), c).detach().cpu().numpy()
                assert np.array_equal(got.reshape(-1), cpu[k].detach().numpy().reshape(-1)), (i, c, k)
    assert abs(float(oms[0].int_scales_object.detach()[0]) - 1.0) > 5e-4


@pytest.mark.parametrize("free_scale", [False, True])
def test_two_hands_bit_equal(free_scale, mano_model):
    """Two hands per frame (right + left, rows interleaved frame-major; reference homan/homan.py:62-63, 341-358, lossutils.py:
    53-59, 116-127), step-2 loss set: the per-hand pair terms (search, contact, interaction records, the three collision scenes),
    the hands' rigid backward as a launch of its own, the MANO backward per hand through its side's model - every stage, every
    parameter gradient, then 6 free-running steps bit-equal in every parameter."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    opts = dict(optimize_object_scale=True) if free_scale else {}
    hm, om = _pair(mano_model, seed=31, frames=6, size=128, obj="bottle", hands=("right", "left"), **opts)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, amp in (("mano_betas", 0.3), ("mano_pca_pose", 0.2), ("mano_trans", 0.01), ("mano_rot", 0.05)):
            d = amp * torch.randn(getattr(om, name).shape, generator=g)
            getattr(om, name).add_(d)
            getattr(hm, name).add_(d.to(getattr(hm, name).device))
        d = 0.5 * (om.translations_object.repeat_interleave(2, 0) - om.translations_hand)     # both hands into the object
        om.translations_hand.add_(d)
        hm.translations_hand.add_(d.to(hm.translations_hand.device))
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    two = handchain.two_hand_terms(om, lw)
    assert np.array_equal(st.vh.cpu().numpy(), two["vh"]) and np.array_equal(st.vm.cpu().numpy(), two["mesh"])
    for n, (arr, _) in zip(("U_smh", "U_v2d", "U_colh", "U_colh2", "U_conh"), two["terms"]):
        assert np.array_equal(getattr(st, n).cpu().numpy(), arr), n
    assert np.abs(two["terms"][2][0]).max() > 0 and np.abs(two["terms"][3][0]).max() > 0        # both collision families live
    assert np.array_equal(st.rec.cpu().numpy()[:, [0, 2, 3, 4]], two["rec"][:, [0, 2, 3, 4]])
    want, stg = handchain.hand_param_grads(om, lw, return_stages=True, two=two)
    assert np.array_equal(st.G_mesh.cpu().numpy(), stg["g_mesh"])
    want.update(objchain.object_pose_grads(om, lw, obj_terms=two["obj_terms"]))
    assert ("int_scales_object" in want) == free_scale
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = _pair(mano_model, seed=32, frames=6, size=128, obj="bottle", hands=("right", "left"), **opts)
    st = FusedStepper(hm, lw, 1e-2, 6)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(6):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)


def _two_hand_depth_pair(mano_model, seed, frames, size):
    """two hands per frame with instance masks that DISAGREE with the initial geometry (the object annotated in front of both
    hands everywhere, the hands moved over it): all three pairs of the three-layer ordinal depth term are live"""
    from homan_amd import HOMan, synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn, hands=("right", "left"))
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
        op["full_mask"] = ((pp["masks"].sum(0) > 0) | (op["full_mask"] > 0)).float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([[[0.05, 0.0, -0.02]], [[-0.05, 0.0, -0.02]]])
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=size,
                  mano_model=mano_model, rend_size=size, ordinal_depth=True)
    return HOMan(**copy.deepcopy(kw), **common), OracleHOMan(**copy.deepcopy(kw), **common)


@pytest.mark.parametrize("weights_name", ["STEP1_LOSS_WEIGHTS", "STEP2_LOSS_WEIGHTS"])
def test_two_hands_with_depth_term_bit_equal(weights_name, mano_model):
    """Two hands per frame AND the ordinal depth term (reference homan/homan.py:384-419 with three layers, lossutils.py:133-169
    over the three pairs with one normaliser): the three pooled depth images, the pairs' counts, every layer's summed gradient
    image, the three vertex gradients, all parameter gradients, then 6 free-running steps bit-equal in every parameter.  With
    the step-2 weights the collision families and the depth term feed the hands together."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import depthchain, handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(getattr(synth, weights_name), lw_depth=1.0)
    hm, om = _two_hand_depth_pair(mano_model, seed=41, frames=6, size=128)
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    dep, stg = depthchain.depth_vertex_grads_layers(om, lw["lw_depth"], return_stages=True)
    for li in range(3):
        assert np.array_equal(st.dl_dep[li].cpu().numpy(), stg["pooled"][li]), li
        assert np.array_equal(st.dl_g[li].cpu().numpy(), stg["g_layer"][li]), li
    counts = [float(st.dp_rec[k][0]) for k in range(3)]
    assert counts == [float(n) for n in stg["npairs"]] and min(counts) > 0                 # (every pair is compared somewhere)
    assert [float(st.dp_up[k]) for k in range(3)] == [float(np.float32(np.float32(lw["lw_depth"]) * s)) for s in stg["shares"]]
    assert np.array_equal(st.G_dep_o.cpu().numpy(), dep[0])
    for i in range(2):
        assert np.abs(dep[1 + i]).max() > 0 and np.array_equal(st.G_dep_h_d[i].cpu().numpy(), dep[1 + i]), i
    two = handchain.two_hand_terms(om, lw)
    want = handchain.hand_param_grads(om, lw, two=two)
    want.update(objchain.object_pose_grads(om, lw, obj_terms=two["obj_terms"]))
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = _two_hand_depth_pair(mano_model, seed=42, frames=6, size=128)
    st = FusedStepper(hm, lw, 1e-2, 6)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(6):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)


def test_fixed_hand_mesh_bit_equal(mano_model):
    """optimize_mano=False (reference homan/homan.py:104-106: the hand mesh is an input, only its rigid pose is optimised),
    step-2 set: gradients of the four pose tensors, then 8 free-running steps, bit-equal."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    from homan_amd import HOMan
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)

    def build(seed):
        clip = synth.make_clip(seed=seed, frames=6, rend_size=128, image_size=128, obj="bottle", silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
        kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
        common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=False, image_size=128,
                      mano_model=mano_model, rend_size=128)
        return HOMan(**copy.deepcopy(kw), **common), OracleHOMan(**copy.deepcopy(kw), **common)

    hm, om = build(41)
    with torch.no_grad():
        d = 0.6 * (om.translations_object - om.translations_hand)
        om.translations_hand.add_(d)
        hm.translations_hand.add_(d.to(hm.translations_hand.device))
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    want, stg = handchain.hand_param_grads(om, lw, return_stages=True)
    assert sorted(want) == ["rotations_hand", "translations_hand"]
    want.update(objchain.object_pose_grads(om, lw, contact_obj=stg["pair"]["con_obj"]))
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = build(42)
    st = FusedStepper(hm, lw, 1e-2, 12)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(8):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)


def test_inter_type_min_with_a_free_object_scale_bit_equal(mano_model):
    """inter_type="min" AND optimize_object_scale: the closest pair's pull then reaches the object as well (its vertex j*, and
    through it pose and scale).  The oracle's written-out chain covers it since round 6 (oracle/objchain.py): all gradients incl.
    the scale's, then 8 free-running steps, bit-equal - step-1 and step-2 sets."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    for lw_name, seed in (("STEP1_LOSS_WEIGHTS", 61), ("STEP2_LOSS_WEIGHTS", 62)):
        lw = dict(getattr(synth, lw_name))
        hm, om = _pair(mano_model, seed=seed, frames=6, size=128, obj="bottle", inter_type="min", optimize_object_scale=True)
        st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
        st.forward_backward(log=True)
        torch.cuda.synchronize()
        want, stg = handchain.hand_param_grads(om, lw, return_stages=True)
        assert np.abs(stg["g_rigid"]).max() > 0
        with torch.no_grad():
            rec = handchain.inter_records(np.ascontiguousarray(om.get_verts_hand()[0].numpy(), np.float32),
                                          np.ascontiguousarray(om.get_verts_object()[0].numpy(), np.float32),
                                          np.ascontiguousarray(om.camintr.numpy(), np.float32))
        pair = stg.get("pair") or {}
        want.update(objchain.object_pose_grads(om, lw, contact_obj=pair.get("con_obj"), inter_rec=rec))
        report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
        assert all(report.values()), (lw_name, report)
        hm, om = _pair(mano_model, seed=seed + 100, frames=6, size=128, obj="bottle", inter_type="min", optimize_object_scale=True)
        st = FusedStepper(hm, lw, 1e-2, 8)
        opt = make_optimizer(om, 1e-2, reproducible=True)
        for i in range(8):
            st.run(1)
            reproducible_step(om, lw, opt)
            torch.cuda.synchronize()
            cpu = dict(om.named_parameters())
            diff = [k for k, p in hm.named_parameters()
                    if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
            assert not diff, (lw_name, i, diff)


def test_inter_type_min_bit_equal(mano_model):
    """inter_type="min" (reference homan/losses.py:219-221, a HOMan option its loop cannot select): the closest hand-object vertex
    pair per gated frame pulls on the hand's rigid pose.  Step-2 set: all gradients, then 8 free-running steps, bit-equal."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle import handchain, objchain
    from oracle.jointopt import make_optimizer, reproducible_step
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    hm, om = _pair(mano_model, seed=51, frames=6, size=128, obj="bottle", inter_type="min")
    st = FusedStepper(hm, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    want, stg = handchain.hand_param_grads(om, lw, return_stages=True)
    assert np.abs(stg["g_rigid"]).max() > 0                                  # (the gate lets frames through: the term is live)
    assert np.array_equal(st.G_min_h.cpu().numpy(), stg["g_rigid"])
    want.update(objchain.object_pose_grads(om, lw, contact_obj=stg["pair"]["con_obj"]))
    report = {k: bool(np.array_equal(getattr(st.model, k).grad.cpu().numpy().reshape(v.shape), v)) for k, v in want.items()}
    assert all(report.values()), report
    hm, om = _pair(mano_model, seed=52, frames=6, size=128, obj="bottle", inter_type="min")
    st = FusedStepper(hm, lw, 1e-2, 12)
    opt = make_optimizer(om, 1e-2, reproducible=True)
    for i in range(8):
        st.run(1)
        reproducible_step(om, lw, opt)
        torch.cuda.synchronize()
        cpu = dict(om.named_parameters())
        diff = [k for k, p in hm.named_parameters()
                if k in cpu and not np.array_equal(p.detach().cpu().numpy(), cpu[k].detach().numpy().reshape(p.shape))]
        assert not diff, (i, diff)
