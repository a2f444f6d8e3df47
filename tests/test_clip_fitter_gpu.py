"""A stream of clips through resident steppers (homan_amd.jointopt.ClipFitter: the sample loop of reference
fit_vid_dataset.py:190-379 without rebuilding model / workspaces / hipGraph for a clip whose shapes have been seen): every
clip ends with exactly the result of a fresh `optimize_hand_object` fit, bit for bit, and the process holds a bounded number
of graphs."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
STEPS = 12


def _clip(mano, seed, frames=4, size=64, obj="cube"):
    from homan_amd import synth
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    return synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)


def _fresh(mano, clip, lw, size=64):
    from homan_amd.jointopt import optimize_hand_object
    model, evo, _ = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                         objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                         loss_weights=lw, num_iterations=STEPS, optimize_mano=True, image_size=size,
                                         mano_model=mano, rend_size=size)
    return model, evo


def _same(result, model, evo):
    for k, v in result["state_dict"].items():
        assert torch.equal(v, getattr(model, k).detach().cpu()), k
    assert torch.equal(result["verts_object"], model.get_verts_object()[0].detach().cpu())
    assert torch.equal(result["verts_hand"], model.get_verts_hand()[0].detach().cpu())
    for k in evo:
        np.testing.assert_array_equal(np.asarray(result["loss_evolution"][k]), np.asarray(evo[k]), err_msg=k)


@pytest.mark.parametrize("step2", [False, True])
def test_reloaded_stepper_equals_fresh_fits(step2, mano_model):
    from homan_amd import lib, synth
    from homan_amd.jointopt import ClipFitter
    lw = dict(synth.STEP2_LOSS_WEIGHTS if step2 else synth.STEP1_LOSS_WEIGHTS)
    clips = [_clip(mano_model, s) for s in (11, 12, 13)]
    fitter = ClipFitter(lw, num_iterations=STEPS, optimize_mano=True, image_size=64, mano_model=mano_model, rend_size=64)
    graphs0 = len(lib._KEPT_GRAPHS)
    results = fitter.fit(clips)
    assert fitter.timing["built"] == 1 and fitter.timing["reused"] == 2
    # one resident stepper for the three clips: its one-iteration graph and its K-iterations-per-replay twin, nothing per clip
    assert len(lib._KEPT_GRAPHS) - graphs0 == 2
    for clip, res in zip(clips, results):
        model, evo = _fresh(mano_model, clip, lw)
        _same(res, model, evo)


def test_mixed_shapes_batches_and_eviction(mano_model):
    """cube and bottle clips interleaved: one resident stepper per shape; two clips per batch; with room for ONE resident
    shape the steppers are rebuilt on every change of shape and the results are the same"""
    from homan_amd import synth
    from homan_amd.jointopt import ClipFitter
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    clips = [_clip(mano_model, 21, obj="cube"), _clip(mano_model, 22, obj="bottle"), _clip(mano_model, 23, obj="cube"),
             _clip(mano_model, 24, obj="bottle"), _clip(mano_model, 25, obj="cube"), _clip(mano_model, 26, obj="bottle"),
             _clip(mano_model, 27, obj="cube"), _clip(mano_model, 28, obj="bottle")]
    kw = dict(num_iterations=STEPS, optimize_mano=True, image_size=64, mano_model=mano_model, rend_size=64)
    fresh = [_fresh(mano_model, c, lw) for c in clips]
    one = ClipFitter(lw, **kw)
    for res, (model, evo) in zip(one.fit(clips), fresh):
        _same(res, model, evo)
    assert one.timing["built"] == 2 and one.timing["reused"] == 6
    two = ClipFitter(lw, clips_per_batch=2, **kw)
    for res, (model, evo) in zip(two.fit(clips), fresh):
        _same(res, model, evo)
    assert two.timing["built"] == 2 and two.timing["reused"] == 2
    # a second stream through the same fitter: nothing is built any more
    for res, (model, evo) in zip(two.fit(clips[:4]), fresh[:4]):
        _same(res, model, evo)
    assert two.timing["built"] == 2
    small = ClipFitter(lw, max_resident=1, **kw)
    for res, (model, evo) in zip(small.fit(clips[:2]) + small.fit(clips[2:4]), fresh[:4]):
        _same(res, model, evo)
    assert small.timing["built"] == 4 and len(small.resident) == 1


def test_clip_without_intrinsics_and_one_at_a_time_configurations(mano_model):
    """ADVICE r4.  (1) A clip WITHOUT `camintr` behind one with: the resident model takes the constructor's default intrinsics
    (reference homan.py:113-116), not the previous clip's - bit-identical to a fresh fit.  (2) Configurations the fused loop
    takes one clip at a time (two hands per frame) with clips_per_batch=2: fitted clip by clip through a resident one-clip
    stepper instead of raising out of fit(), results those of fresh fits."""
    from homan_amd import synth
    from homan_amd.jointopt import ClipFitter
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    a, b = _clip(mano_model, 31), _clip(mano_model, 32)
    b = dict(b, camintr=None)
    fitter = ClipFitter(lw, num_iterations=STEPS, optimize_mano=True, image_size=64, mano_model=mano_model, rend_size=64)
    res = fitter.fit([a, b])
    assert fitter.timing["built"] == 1 and fitter.timing["reused"] == 1
    for clip, r in zip((a, b), res):
        _same(r, *_fresh(mano_model, clip, lw))
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    two = [synth.make_clip(seed=s, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn, hand_verts_fn=hand_fn,
                           hands=("right", "left")) for s in (41, 42, 43)]
    pair = ClipFitter(lw, num_iterations=STEPS, clips_per_batch=2, optimize_mano=True, image_size=64, mano_model=mano_model,
                      rend_size=64)
    out = pair.fit(two)
    assert len(out) == 3 and pair.timing["built"] == 1 and pair.timing["reused"] == 2
    for clip, r in zip(two, out):
        _same(r, *_fresh(mano_model, clip, lw))


def test_depth_term_in_resident_clip_batches(mano_model):
    """The ordinal depth term (opt-in, reference homan.py:384-419) through ClipFitter with two clips per batch: the second
    batch is COPIED into the resident stepper - its instance masks included (round 5: a resident batch used to refuse the
    reload and the fitter fell back to one clip per stepper) - and every clip equals its fresh solo fit bit for bit."""
    from homan_amd import synth
    from homan_amd.jointopt import ClipFitter, optimize_hand_object
    lw = dict(synth.STEP1_LOSS_WEIGHTS, lw_depth=1.0)
    clips = []
    for s in (51, 52, 53, 54):
        clip = _clip(mano_model, s)
        for pp, op in zip(clip["person_parameters"], clip["object_parameters"]):      # annotations that disagree with the geometry
            op["full_mask"] = ((pp["masks"][0] > 0) | (op["full_mask"] > 0)).float()
            pp["masks"] = torch.zeros_like(pp["masks"])
            pp["translations"] = pp["translations"] + torch.tensor([0.06, 0.0, -0.02])
        clips.append(clip)
    fitter = ClipFitter(lw, num_iterations=STEPS, clips_per_batch=2, optimize_mano=True, image_size=64, mano_model=mano_model,
                        rend_size=64, ordinal_depth=True)
    res = fitter.fit(clips)
    assert fitter.timing["built"] == 1 and fitter.timing["reused"] == 1
    for clip, r in zip(clips, res):
        model, evo, _ = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                             objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                             loss_weights=lw, num_iterations=STEPS, optimize_mano=True, image_size=64,
                                             mano_model=mano_model, rend_size=64, ordinal_depth=True)
        assert evo["loss_depth"][0] > 0
        _same(r, model, evo)
