"""Clip batches (BASELINE cfg4 / cfg5 in miniature): C clips optimised by ONE launch per kernel must equal C
single-clip optimisations BIT FOR BIT - loss_evolution rows and final parameters - and the shared object scale of
cfg5 must follow the tied-parameter semantics of homan_amd.dist.  GPU box."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clips(mano, seeds, frames, size, obj, **kw):
    from homan_amd import synth
    from homan_amd.jointopt import build_model
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    models = []
    for seed in seeds:
        clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
        models.append(build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                  objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                  optimize_mano=True, image_size=size, mano_model=mano, rend_size=size,
                                  sync_metrics=False, **kw))
    return models


PARAMS = ["translations_object", "rotations_object", "translations_hand", "rotations_hand", "mano_pca_pose", "mano_betas",
          "mano_rot", "mano_trans", "int_scales_object"]


def _compare(mano, seeds, frames, size, obj, lw, steps, **kw):
    from homan_amd.jointopt import FusedStepper
    singles = _clips(mano, seeds, frames, size, obj, **kw)
    evo_single = []
    for m in singles:
        st = FusedStepper(m, lw, 1e-2, steps)
        st.run(steps)
        evo_single.append(st.loss_evolution(steps))
    batch = _clips(mano, seeds, frames, size, obj, **kw)
    st = FusedStepper(batch, lw, 1e-2, steps)
    st.run(steps)
    evo_batch = st.loss_evolution(steps)
    assert len(evo_batch) == len(seeds)
    for c, (es, eb) in enumerate(zip(evo_single, evo_batch)):
        assert sorted(es) == sorted(eb)
        for k in es:
            np.testing.assert_array_equal(np.asarray(eb[k]), np.asarray(es[k]), err_msg=f"clip {c} {k}")
        assert np.isfinite(es["loss"]).all() and min(es["loss"]) > 0
    for c, (ms, mb) in enumerate(zip(singles, batch)):
        for k in PARAMS:
            a, b = getattr(ms, k).detach(), getattr(mb, k).detach()
            assert torch.equal(a, b), f"clip {c} {k}: max diff {(a - b).abs().max().item()}"
        # each model of the batch holds its own result: its own forward sees the optimised parameters
        assert torch.equal(ms.get_verts_object()[0], mb.get_verts_object()[0])


@pytest.mark.parametrize("step2", [False, True])
def test_batched_step_equals_single_steps_bitwise(mano_model, step2):
    from homan_amd import synth
    lw = dict(synth.STEP2_LOSS_WEIGHTS if step2 else synth.STEP1_LOSS_WEIGHTS)
    _compare(mano_model, [3, 4, 5], 4, 64, "cube", lw, 12)


def test_batched_free_object_scale_bitwise(mano_model):
    from homan_amd import synth
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    lw["lw_scale_obj"] = 10.0
    _compare(mano_model, [11, 12], 5, 64, "cube", lw, 8, optimize_object_scale=True)


def test_cfg1_weights_batched_bitwise(mano_model):
    """BASELINE cfg1's loss set (silhouette + 2-D keypoints only): the separate v2d / priors launches of the fused loop."""
    from homan_amd import synth
    _compare(mano_model, [0, 1], 10, 128, "cube", dict(synth.CFG1_LOSS_WEIGHTS), 6)


def test_cfg4_miniature_full_size_clips(mano_model):
    """two cfg2-sized clips (30 frames, 256^2, 3000-face bottle) in one batch == alone, bit for bit"""
    from homan_amd import synth
    _compare(mano_model, [0, 1], 30, 256, "bottle", dict(synth.STEP1_LOSS_WEIGHTS), 5)


def test_shared_scale_fused_matches_tied_eager(mano_model):
    """cfg5 semantics on one rank (process group of size 1, backend nccl = RCCL): the fused loop with `shared_scale`
    keeps the replicas of the scalar identical and follows the eager tied-parameter loop of homan_amd.dist."""
    import torch.distributed as dist
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, parameter_groups
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        lw = dict(synth.STEP2_LOSS_WEIGHTS)
        lw["lw_scale_obj"] = 10.0
        steps = 4
        eager = _clips(mano_model, [21, 22], 4, 64, "cube", optimize_object_scale=True)
        opts = [torch.optim.Adam(parameter_groups(m, 1e-2)) for m in eager]
        hist = hdist.optimize_clips_shared_scale(eager, opts, lw, steps)
        fused = _clips(mano_model, [21, 22], 4, 64, "cube", optimize_object_scale=True)
        st = FusedStepper(fused, lw, 1e-2, steps, shared_scale=True)
        st.run(steps)
        evo = st.loss_evolution(steps)
        s = st.model.int_scales_object.detach().cpu().numpy()
        assert s[0] == s[1] and abs(float(s[0]) - 1.0) > 1e-4          # replicas identical, and the scalar moved
        np.testing.assert_allclose(s[0], eager[0].int_scales_object.detach().cpu().numpy()[0], rtol=2e-4)
        np.testing.assert_allclose([evo[0]["loss"][0], evo[1]["loss"][0]], hist[0], rtol=2e-4)
        np.testing.assert_allclose([evo[0]["loss"][1], evo[1]["loss"][1]], hist[1], rtol=5e-3)
    finally:
        if created:
            dist.destroy_process_group()


def test_heterogeneous_shard_equals_solo_runs_bitwise(mano_model):
    """A shard as a real dataset hands it over (reference homan/datasets/core50.py:22-42: every clip its own object mesh;
    fit_vid_dataset.py:190: its own length): cube clips of 8 frames and a bottle clip of 10, in arbitrary order.  ShardStepper
    batches the clips that agree in shape and runs the batches of different shapes one after the other inside every iteration;
    every clip ends up with exactly the rows and parameters of optimising it alone."""
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, ShardStepper
    lw, steps = dict(synth.STEP2_LOSS_WEIGHTS), 6
    spec = [(3, 8, "cube"), (4, 10, "bottle"), (5, 8, "cube")]
    build = lambda: [_clips(mano_model, [seed], frames, 64, obj)[0] for seed, frames, obj in spec]
    solo, evo_solo = build(), []
    for m in solo:
        st = FusedStepper(m, lw, 1e-2, steps)
        st.run(steps)
        evo_solo.append(st.loss_evolution(steps))
    shard = build()
    sh = ShardStepper(shard, lw, 1e-2, steps)
    assert sorted(len(i) for i in sh.index) == [1, 2]              # the two cubes share a batch, the bottle has its own
    sh.run(steps)
    for c, (es, eb) in enumerate(zip(evo_solo, sh.loss_evolution(steps))):
        for k in es:
            np.testing.assert_array_equal(np.asarray(eb[k]), np.asarray(es[k]), err_msg=f"clip {c} {k}")
    for c, (ms, mb) in enumerate(zip(solo, shard)):
        for k in PARAMS:
            assert torch.equal(getattr(ms, k).detach(), getattr(mb, k).detach()), f"clip {c} {k}"
    # the public entry point (what a rank of bench.py --gpus N / a dataset driver calls) takes the same shard
    again = build()
    evo = hdist.optimize_clip_shard(again, lw, steps)
    for es, eb in zip(evo_solo, evo):
        np.testing.assert_array_equal(np.asarray(eb["loss"]), np.asarray(es["loss"]))


def test_heterogeneous_shard_with_a_tied_scale(mano_model):
    """cfg5 semantics over clips of different shapes: one scalar for all of them, replicas identical in every shape group,
    and the same trajectory as the plain autograd tied loop of homan_amd.dist over the same (HIP) models."""
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import ShardStepper, parameter_groups
    lw, steps = dict(synth.STEP2_LOSS_WEIGHTS), 4
    lw["lw_scale_obj"] = 10.0
    spec = [(3, 8, "cube"), (4, 10, "bottle"), (5, 8, "cube")]
    build = lambda: [_clips(mano_model, [seed], frames, 64, obj, optimize_object_scale=True)[0] for seed, frames, obj in spec]
    eager = build()
    opts = [torch.optim.Adam(parameter_groups(m, 1e-2)) for m in eager]
    hist = hdist.optimize_clips_shared_scale(eager, opts, lw, steps)
    shard = build()
    sh = ShardStepper(shard, lw, 1e-2, steps, shared_scale=True)
    sh.run(steps)
    s = np.asarray([float(m.int_scales_object.detach().cpu()[0]) for m in shard], np.float32)
    assert (s == s[0]).all() and abs(float(s[0]) - 1.0) > 1e-4
    np.testing.assert_allclose(s[0], float(eager[0].int_scales_object.detach().cpu()[0]), rtol=2e-4)
    evo = sh.loss_evolution(steps)
    np.testing.assert_allclose([e["loss"][0] for e in evo], hist[0], rtol=2e-4)


def test_cfg4_full_shard_eight_full_size_clips(mano_model):
    """BASELINE cfg4 as one rank of it runs: EIGHT clips of 30 frames at 256^2 (bottle, step-1 losses) as one clip batch - one
    launch per kernel over 240 frames.  Every clip's rows are finite, and clip 5 (any one would do) is bit-identical to the
    same clip optimised alone."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    lw, steps = dict(synth.STEP1_LOSS_WEIGHTS), 4
    batch = _clips(mano_model, list(range(8)), 30, 256, "bottle")
    st = FusedStepper(batch, lw, 1e-2, steps)
    st.run(steps)
    evo = st.loss_evolution(steps)
    assert len(evo) == 8
    for e in evo:
        assert all(np.isfinite(v).all() for v in e.values())
    (alone,) = _clips(mano_model, [5], 30, 256, "bottle")
    sa = FusedStepper(alone, lw, 1e-2, steps)
    sa.run(steps)
    for k, v in sa.loss_evolution(steps).items():
        np.testing.assert_array_equal(np.asarray(evo[5][k]), np.asarray(v), err_msg=k)
    for k in PARAMS:
        if hasattr(alone, k):
            assert torch.equal(getattr(alone, k).detach(), getattr(batch[5], k).detach()), k


def test_shard_of_two_hand_clips_runs_through_the_fused_loop(mano_model):
    """A batch of two-hand clips is not ONE launch per kernel (the fused loop takes two hands per frame one clip at a time,
    reference homan.py:341-358): `ShardStepper` gives every such clip a stepper of its own and replays them side by side - each
    clip ends exactly where a solo fit ends."""
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, ShardStepper, build_model
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    lw = dict(synth.STEP2_LOSS_WEIGHTS)

    def models():
        out = []
        for s in (31, 32, 33):
            c = synth.make_clip(seed=s, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn,
                                hand_verts_fn=hand_fn, hands=("right", "left"))
            out.append(build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                   objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True,
                                   image_size=64, mano_model=mano_model, rend_size=64, sync_metrics=False))
        return out
    shard = models()
    evo = hdist.optimize_clip_shard(shard, lw, 6)
    solo = models()
    for m_s, m_b, e in zip(solo, shard, evo):
        st = FusedStepper(m_s, lw, 1e-2, 6)
        st.run(6)
        for (k, p), (_, q) in zip(m_s.named_parameters(), m_b.named_parameters()):
            assert torch.equal(p, q), k
        np.testing.assert_array_equal(np.asarray(st.loss_evolution(6)["loss"]), np.asarray(e["loss"]))


def test_one_shape_shard_runs_as_two_batches_and_equals_solo_runs(mano_model, monkeypatch):
    """A step-1 shard of ONE shape: ShardStepper fits it as two clip batches side by side (five clips: 3 + 2); every clip ends up
    with the rows and parameters of optimising it alone, and with those of the one-batch run (HOMAN_SHARD_SPLIT=0).  With the
    collision / contact terms the shard stays one batch."""
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, ShardStepper
    lw, steps, seeds = dict(synth.STEP1_LOSS_WEIGHTS), 6, [11, 12, 13, 14, 15]
    solo, evo_solo = _clips(mano_model, seeds, 6, 64, "cube"), []
    for m in solo:
        st = FusedStepper(m, lw, 1e-2, steps)
        st.run(steps)
        evo_solo.append(st.loss_evolution(steps))
    shard = _clips(mano_model, seeds, 6, 64, "cube")
    sh = ShardStepper(shard, lw, 1e-2, steps)
    assert [len(i) for i in sh.index] == [3, 2]
    sh.run(steps)
    monkeypatch.setenv("HOMAN_SHARD_SPLIT", "0")
    whole = _clips(mano_model, seeds, 6, 64, "cube")
    sw = ShardStepper(whole, lw, 1e-2, steps)
    assert [len(i) for i in sw.index] == [5]
    sw.run(steps)
    monkeypatch.delenv("HOMAN_SHARD_SPLIT")
    for c, (es, eb, ew) in enumerate(zip(evo_solo, sh.loss_evolution(steps), sw.loss_evolution(steps))):
        for k in es:
            np.testing.assert_array_equal(np.asarray(eb[k]), np.asarray(es[k]), err_msg=f"clip {c} {k}")
            np.testing.assert_array_equal(np.asarray(ew[k]), np.asarray(es[k]), err_msg=f"clip {c} {k} (one batch)")
    for c, (ms, mb) in enumerate(zip(solo, shard)):
        for k in PARAMS:
            assert torch.equal(getattr(ms, k).detach(), getattr(mb, k).detach()), f"clip {c} {k}"
    assert [len(i) for i in ShardStepper(_clips(mano_model, seeds[:4], 6, 64, "cube"), dict(synth.STEP2_LOSS_WEIGHTS), 1e-2, 2,
                                         capture=False).index] == [4]
