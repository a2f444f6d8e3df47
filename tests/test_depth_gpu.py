"""Ordinal-depth row on the GPU: depth image, its backward, the loss and the model term vs the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(mano_model, frames=4, size=64):
    from homan_amd import synth
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=11, frames=frames, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    K = torch.from_numpy(clip["camintr"]).float()
    return clip, K


def _oracle_render(verts, faces, K, size):
    from oracle import nmr
    r = nmr.Renderer(image_size=size, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    _, depth, alpha = r.render(verts, faces)
    return alpha, depth


@pytest.mark.parametrize("which", ["object", "hand"])
def test_depth_image_and_backward_match_oracle(which, mano_model):
    from homan_amd import ops
    size = 64
    clip, K = _scene(mano_model, size=size)
    if which == "object":
        verts, faces = clip["gt"]["verts_object"], clip["objfaces"]
    else:
        verts = clip["gt"]["verts_hand"]
        faces = torch.from_numpy(mano_model["faces"].astype(np.int32))[None].repeat(verts.shape[0], 1, 1)
    B, V = verts.shape[:2]
    vo = verts.clone().requires_grad_(True)
    a_o, d_o = _oracle_render(vo, faces, K, size)
    g = torch.from_numpy(np.random.default_rng(3).normal(size=tuple(d_o.shape)).astype(np.float32))
    (d_o * g).sum().backward()

    sctx = ops.SilhouetteContext(faces.to(DEV), V, B, size, DEV)
    vh = verts.clone().to(DEV).requires_grad_(True)
    a_h, d_h = ops.depth_render(vh, K.to(DEV), sctx, 1.0)
    np.testing.assert_array_equal(a_h.cpu().numpy(), a_o.detach().numpy())
    np.testing.assert_array_equal(d_h.detach().cpu().numpy(), d_o.detach().numpy())      # same bits
    (d_h * g.to(DEV)).sum().backward()
    want, got = vo.grad.numpy(), vh.grad.cpu().numpy()
    scale = np.abs(want).max()
    assert scale > 0
    np.testing.assert_allclose(got / scale, want / scale, atol=1e-4)


@pytest.mark.parametrize("pattern", ["none", "one_pixel", "band", "all"])
def test_sparse_depth_backward_equals_the_dense_one(pattern, mano_model):
    """hm_depth_bwd_sparse with the non-zero flags of the upstream image (as hm_ordinal_depth_bwd_flags writes them) returns the
    vertex gradients of hm_depth_bwd - faces and frames that touch no flagged segment are skipped, not approximated -, for an
    all-zero image, one pixel, a band of rows in some frames and a dense image; and the flags kernel marks exactly the segments
    that hold a non-zero gradient."""
    from homan_amd import lib as hlib, ops
    size = 128
    clip, K = _scene(mano_model, frames=5, size=size)
    verts, faces = clip["gt"]["verts_object"].to(DEV).contiguous(), clip["objfaces"]
    B, V = verts.shape[:2]
    sctx = ops.SilhouetteContext(faces.to(DEV), V, B, size, DEV)
    Kd = K.to(DEV).contiguous()
    with torch.no_grad():
        _, depth = ops.depth_render(verts, Kd, sctx, 1.0)
    rng = np.random.default_rng(5)
    g = np.zeros((B, size, size), np.float32)
    if pattern == "one_pixel":
        ys, xs = np.nonzero(np.isfinite(depth[2].cpu().numpy()) & (depth[2].cpu().numpy() < 50))
        g[2, ys[len(ys) // 2], xs[len(xs) // 2]] = 0.7
    elif pattern == "band":
        g[1, 40:70] = rng.normal(size=(30, size))
        g[3, 60:62, 64:] = rng.normal(size=(2, size - 64))
    elif pattern == "all":
        g[:] = rng.normal(size=g.shape)
    g = torch.from_numpy(g).to(DEV)
    flags = (g.reshape(B, size, size // 64, 64) != 0).any(-1).to(torch.uint8).contiguous()
    L, P = hlib.lib(), hlib.ptr
    out = [torch.full((B, V, 3), 7.0, device=DEV) for _ in range(2)]
    for o, fl in zip(out, (None, flags)):
        hlib.check(L.hm_depth_bwd_sparse(P(verts), P(Kd), B, V, sctx.F, sctx.S, 1.0, P(g), P(sctx.adj_off), P(sctx.adj_items),
                                         P(o), P(fl) if fl is not None else None, P(sctx.workspace), hlib.stream()), "depth bwd")
    assert torch.equal(out[0], out[1])
    assert (out[0] != 0).any() == (pattern != "none")
    if pattern == "band":
        assert not out[0][0].any() and not out[0][4].any() and out[0][1].any()
    # the flags kernel: two layers whose order contradicts the annotation on a few pixels
    d0 = torch.full((B, size, size), 1.0, device=DEV)
    d1 = torch.full((B, size, size), 1.5, device=DEV)
    a = torch.ones(B, size, size, device=DEV)
    m0 = torch.zeros(B, size, size, dtype=torch.uint8, device=DEV)
    m1 = torch.zeros_like(m0)
    m1[1, 10:12, 70:75] = 1                       # annotated: layer 1 in front; rendered: layer 0 in front -> a gradient there
    rec = torch.tensor([float(B), 0.0, 0.0, 10.0, 1.0], device=DEV)
    up = torch.ones(1, device=DEV)
    g0, g1 = torch.empty_like(d0), torch.empty_like(d0)
    f0 = torch.full((B, size, size // 64), 9, dtype=torch.uint8, device=DEV)
    f1 = torch.full_like(f0, 9)
    hlib.check(L.hm_ordinal_depth_bwd_flags(P(d0), P(d1), P(a), P(a), P(m0), P(m1), B, size, P(rec), P(up), P(g0), P(g1), P(f0),
                                            P(f1), hlib.stream()), "ordinal bwd flags")
    for gi, fi in ((g0, f0), (g1, f1)):
        assert torch.equal(fi, (gi.reshape(B, size, size // 64, 64) != 0).any(-1).to(torch.uint8))
    assert int(f0.sum()) == 2 and int(f1.sum()) == 2


def test_ordinal_depth_loss_matches_oracle(mano_model):
    from homan_amd import ops
    from oracle import model as o_model
    size = 64
    clip, K = _scene(mano_model, size=size)
    B = clip["gt"]["verts_object"].shape[0]
    hf = torch.from_numpy(mano_model["faces"].astype(np.int32))[None].repeat(B, 1, 1)
    a0, d0 = _oracle_render(clip["gt"]["verts_object"], clip["objfaces"], K, size)
    # push the hand a little towards the object so that the two layers overlap with mixed ordering
    vh = clip["gt"]["verts_hand"] + torch.tensor([0.05, 0.0, 0.03])
    a1, d1 = _oracle_render(vh, hf, K, size)
    m0 = (a0 == 1)
    m1 = (a1 == 1) & ~m0                                   # annotation: the object owns the overlap
    m1[0] = (a1[0] == 1)                                   # ... except in frame 0: the hand does
    m0[0] = (a0[0] == 1) & ~m1[0]
    d0o, d1o = d0.clone().requires_grad_(True), d1.clone().requires_grad_(True)
    want = o_model.compute_ordinal_depth_loss(torch.stack([m0, m1], 1), [a0 == 1, a1 == 1], [d0o, d1o])["loss_depth"]
    assert float(want) > 0
    want.backward()

    rws = ops.ReduceWorkspace(DEV)
    d0h, d1h = d0.to(DEV).requires_grad_(True), d1.to(DEV).requires_grad_(True)
    got = ops.ordinal_depth_loss(d0h, d1h, a0.to(DEV), a1.to(DEV), m0.to(torch.uint8).to(DEV).contiguous(),
                                 m1.to(torch.uint8).to(DEV).contiguous(), rws)
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-5)
    (got * 1.0).backward()
    for gh, go in ((d0h.grad, d0o.grad), (d1h.grad, d1o.grad)):
        np.testing.assert_allclose(gh.cpu().numpy(), go.numpy(), rtol=1e-4, atol=1e-9)
    # second call through the same workspace (self-resetting ticket)
    got2 = ops.ordinal_depth_loss(d0h.detach(), d1h.detach(), a0.to(DEV), a1.to(DEV),
                                  m0.to(torch.uint8).to(DEV).contiguous(), m1.to(torch.uint8).to(DEV).contiguous(), rws)
    assert got2.item() == got.item()


def _graph_depth_steps(rank, kw, common, lw, out):
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    from homan_amd import HOMan
    from homan_amd.jointopt import GraphStepper
    hm2 = HOMan(**copy.deepcopy(kw), ordinal_depth=True, **common)
    st = GraphStepper(hm2, lw, 1e-2, 20)
    st.run(20)
    np.save(out, np.asarray(st.loss_evolution(20)["loss"]))


def test_model_ordinal_depth_term_matches_oracle(mano_model, tmp_path):
    """HOMan(ordinal_depth=True) vs OracleHOMan(ordinal_depth=True): loss_depth and its parameter gradients;
    without the opt-in both reproduce the reference's TypeError (homan.py:506-507)."""
    from homan_amd import HOMan, synth
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    size = 64
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=5, frames=4, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    # instance masks that disagree with the initial geometry: the object is annotated in front everywhere
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
        full = ((pp["masks"][0] > 0) | (op["full_mask"] > 0))
        op["full_mask"] = full.float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([0.06, 0.0, -0.02])    # hand over the object
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                  image_size=size, mano_model=mano_model, rend_size=size)
    lw = dict({k: 0.0 for k in synth.STEP1_LOSS_WEIGHTS}, lw_depth=1.0)
    with pytest.raises(TypeError):
        HOMan(**copy.deepcopy(kw), **common)(loss_weights=lw)
    om = OracleHOMan(**copy.deepcopy(kw), ordinal_depth=True, **common)
    hm = HOMan(**copy.deepcopy(kw), ordinal_depth=True, **common)
    lo, _ = om(loss_weights=lw)
    lh, _ = hm(loss_weights=lw)
    assert sorted(lo) == sorted(lh) == ["loss_depth"]
    assert float(lo["loss_depth"]) > 0
    np.testing.assert_allclose(lh["loss_depth"].item(), lo["loss_depth"].item(), rtol=1e-4)
    lo["loss_depth"].backward()
    lh["loss_depth"].backward()
    go = {k: p.grad for k, p in om.named_parameters() if p.grad is not None}
    gh = {k: p.grad for k, p in hm.named_parameters() if p.grad is not None}
    assert sorted(go) == sorted(gh) and len(go) > 0
    for k in go:
        scale = max(go[k].abs().max().item(), 1e-12)
        np.testing.assert_allclose(gh[k].cpu().numpy() / scale, go[k].numpy() / scale, atol=1e-3, err_msg=k)
    # a few optimisation steps on the depth term alone reduce it (graph loop over HOMan.forward + autograd).  In a process of
    # its own: a captured autograd iteration with the depth renders in it, followed later IN THE SAME PROCESS by a
    # three-stream clip-batch graph, has crashed inside the HIP graph runtime at that later replay (measured: this test +
    # test_fused_loop_at_a_render_size_off_the_tile_grid, nothing else needed); neither graph is at fault on its own.
    import torch.multiprocessing as mp
    out = tmp_path / "evo.npy"
    mp.spawn(_graph_depth_steps, args=(kw, common, lw, str(out)), nprocs=1, join=True)
    evo = np.load(out)
    assert evo[-1] < evo[0]


def test_fused_loop_covers_the_ordinal_depth_term(mano_model):
    """FusedStepper with lw_depth > 0 (opt-in `ordinal_depth=True`): its launch sequence - two depth renders at the image
    camera, the ordinal loss, its backward through both depth images into the rigid / MANO backward - gives the losses and the
    parameter gradients of HOMan.forward + autograd, next to the other step-2 terms; without the opt-in it raises the
    reference's TypeError; a few captured steps reduce the term."""
    from homan_amd import HOMan, synth
    from homan_amd.jointopt import FusedStepper
    from oracle.jointopt import collate_inputs
    size = 64
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    clip = synth.make_clip(seed=5, frames=4, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
        full = ((pp["masks"][0] > 0) | (op["full_mask"] > 0))
        op["full_mask"] = full.float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([0.06, 0.0, -0.02])    # hand over the object
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=size,
                  mano_model=mano_model, rend_size=size, sync_metrics=False)
    lw = dict(synth.STEP2_LOSS_WEIGHTS, lw_depth=2.0)
    with pytest.raises(TypeError):
        FusedStepper(HOMan(**copy.deepcopy(kw), **common), lw, 1e-2, 2, capture=False)
    model = HOMan(**copy.deepcopy(kw), ordinal_depth=True, **common)
    loss_dict, _ = model(loss_weights=lw)
    assert float(loss_dict["loss_depth"]) > 0
    total = sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict)
    total.sum().backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    ref_losses = {k: float(v.detach().reshape(-1)[0]) for k, v in loss_dict.items()}
    st = FusedStepper(model, lw, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    for k, v in ref_losses.items():
        np.testing.assert_allclose(st.log_buf[0, 0, st.SLOTS.index(k)].item(), v, rtol=2e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(st.log_buf[0, 0, len(st.SLOTS)].item(), float(total.detach().reshape(-1)[0]), rtol=2e-6)
    for k, p in model.named_parameters():
        if k in ref_grads:
            scale = max(ref_grads[k].abs().max().item(), 1e-20)
            assert ((p.grad - ref_grads[k]).abs().max() / scale).item() < 2e-5, k
    # captured loop on the depth term alone: it goes down
    model2 = HOMan(**copy.deepcopy(kw), ordinal_depth=True, **common)
    lw_d = dict({k: 0.0 for k in synth.STEP1_LOSS_WEIGHTS}, lw_depth=1.0)
    st2 = FusedStepper(model2, lw_d, 1e-2, 20)
    st2.run(20)
    evo = st2.loss_evolution(20)
    assert sorted(evo) == ["loss", "loss_depth"] and evo["loss_depth"][-1] < evo["loss_depth"][0]


def test_depth_term_in_a_clip_batch_equals_solo_runs_bitwise(mano_model):
    """The ordinal depth term normalises over ONE clip (pairs, mask counts): in a clip batch the two depth renders run over
    all frames at once and the term per clip on its slice - rows and parameters bit-identical to optimising the clips alone."""
    from homan_amd import HOMan, synth
    from homan_amd.jointopt import FusedStepper
    from oracle.jointopt import collate_inputs
    size, steps = 64, 5
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    lw = dict(synth.STEP2_LOSS_WEIGHTS, lw_depth=2.0)

    def make(seed):
        clip = synth.make_clip(seed=seed, frames=4, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
        for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
            op["full_mask"] = ((pp["masks"][0] > 0) | (op["full_mask"] > 0)).float()
            pp["masks"] = torch.zeros_like(pp["masks"])
            pp["translations"] = pp["translations"] + torch.tensor([0.06, 0.0, -0.02])    # hand over the object
        kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
        return HOMan(**kw, camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                     image_size=size, mano_model=mano_model, rend_size=size, sync_metrics=False, ordinal_depth=True)
    solo = []
    for seed in (5, 6):
        m = make(seed)
        st = FusedStepper(m, lw, 1e-2, steps)
        st.run(steps)
        solo.append((m, st.loss_evolution(steps)))
        assert max(solo[-1][1]["loss_depth"]) > 0
    batch = [make(5), make(6)]
    sb = FusedStepper(batch, lw, 1e-2, steps)
    sb.run(steps)
    for (ms, es), mb, eb in zip(solo, batch, sb.loss_evolution(steps)):
        for k in es:
            np.testing.assert_array_equal(np.asarray(eb[k]), np.asarray(es[k]), err_msg=k)
        for k in ("translations_object", "rotations_object", "translations_hand", "rotations_hand", "mano_pca_pose"):
            assert torch.equal(getattr(ms, k).detach(), getattr(mb, k).detach()), k


def test_ordinal_depth_term_with_two_hands_matches_oracle(mano_model):
    """hand_nb = 2: the three layers [object, right hand, left hand] of reference homan.py:384-419 pair-wise (lossutils.py:133-169,
    one normaliser over every ordered pair incl. a layer with itself).  HOMan(ordinal_depth=True) vs the oracle: value and
    parameter gradients; a few eager steps of `optimize_hand_object` bring the term down."""
    from homan_amd import HOMan, synth
    from homan_amd.jointopt import optimize_hand_object
    from oracle.jointopt import collate_inputs
    from oracle.model import OracleHOMan
    size = 64
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=4, frames=4, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn, hands=("right", "left"))
    # annotations that disagree with the geometry: the object in front everywhere, both hands moved over it
    for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):
        op["full_mask"] = ((pp["masks"].sum(0) > 0) | (op["full_mask"] > 0)).float()
        pp["masks"] = torch.zeros_like(pp["masks"])
        pp["translations"] = pp["translations"] + torch.tensor([[[0.05, 0.0, -0.02]], [[-0.05, 0.0, -0.02]]])[:pp["translations"].shape[0]]
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    common = dict(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=size,
                  mano_model=mano_model, rend_size=size, ordinal_depth=True)
    lw = dict({k: 0.0 for k in synth.STEP1_LOSS_WEIGHTS}, lw_depth=1.0)
    om = OracleHOMan(**copy.deepcopy(kw), **common)
    hm = HOMan(**copy.deepcopy(kw), **common)
    assert hm.hand_nb == 2
    lo, _ = om(loss_weights=lw)
    lh, _ = hm(loss_weights=lw)
    assert float(lo["loss_depth"]) > 0
    np.testing.assert_allclose(lh["loss_depth"].item(), lo["loss_depth"].item(), rtol=1e-4)
    lo["loss_depth"].backward()
    lh["loss_depth"].backward()
    go = {k: p.grad for k, p in om.named_parameters() if p.grad is not None}
    gh = {k: p.grad for k, p in hm.named_parameters() if p.grad is not None}
    assert sorted(go) == sorted(gh) and len(go) > 0
    for k in go:
        scale = max(go[k].abs().max().item(), 1e-12)
        np.testing.assert_allclose(gh[k].cpu().numpy() / scale, go[k].numpy() / scale, atol=1e-3, err_msg=k)
    model, evo, _ = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                         objvertices=clip["objvertices"], objfaces=clip["objfaces"], loss_weights=lw,
                                         num_iterations=8, lr=1e-2, camintr=clip["camintr"], optimize_mano=True, image_size=size,
                                         mano_model=mano_model, rend_size=size, ordinal_depth=True, mode="eager")
    assert evo["loss_depth"][-1] < evo["loss_depth"][0]
    # the FUSED loop on the same configuration (round 5): three renders, three pair terms, the scene's normaliser and the
    # pairs' shares on the device, one depth-map backward per layer - losses and gradients of HOMan.forward + autograd, next
    # to the other step-2 terms (three collision scenes, per-hand contact), and a few captured steps bring the term down
    from homan_amd.jointopt import FusedStepper
    lw2 = dict(synth.STEP2_LOSS_WEIGHTS, lw_depth=2.0)
    ref = HOMan(**copy.deepcopy(kw), **common, sync_metrics=False)
    ld, md = ref(loss_weights=lw2)
    total = sum(ld[k] * lw2[k.replace("loss", "lw")] for k in ld)
    total.sum().backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
    ref_losses = {k: float(v.detach().reshape(-1)[0]) for k, v in ld.items()}
    assert ref_losses["loss_depth"] > 0
    fm = HOMan(**copy.deepcopy(kw), **common, sync_metrics=False)
    st = FusedStepper(fm, lw2, 1e-2, 4, capture=False)
    st.forward_backward(log=True)
    torch.cuda.synchronize()
    for k, v in ref_losses.items():
        np.testing.assert_allclose(st.log_buf[0, 0, st.SLOTS.index(k)].item(), v, rtol=2e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(st.log_buf[0, 0, len(st.SLOTS)].item(), float(total.detach().reshape(-1)[0]), rtol=2e-6)
    for k, p in fm.named_parameters():
        if k in ref_grads:
            scale = max(ref_grads[k].abs().max().item(), 1e-20)
            assert ((p.grad - ref_grads[k]).abs().max() / scale).item() < 2e-5, k
    fm2 = HOMan(**copy.deepcopy(kw), **common, sync_metrics=False)
    st2 = FusedStepper(fm2, lw, 1e-2, 12)
    st2.run(12)
    evo2 = st2.loss_evolution(12)
    assert evo2["loss_depth"][-1] < evo2["loss_depth"][0]
    np.testing.assert_allclose(evo2["loss_depth"][0], lh["loss_depth"].item(), rtol=2e-6)
