"""N > 1 ranks THROUGH THE PRODUCT PATH on the one GPU a test box has, and BASELINE cfg5 at full size.

* Two (and three) processes share cuda:0 under `gloo` (device tensors staged through the host there, homan_amd.dist._staged):
  each rank runs `homan_amd.dist.optimize_clip_shard(..., shared_scale=True)` - the fused launch sequence replayed from two
  hipGraphs around the collective - on its shard of the clips.  What RCCL would change is the transport of ONE fp32 per step;
  sharding, the graph split around the collective, the replica bookkeeping and the empty-rank protocol are the code under test.
* cfg5 itself: 8 clips x 30 frames x 256^2, step-2 losses, one tied object scale, one-rank nccl (= RCCL) group.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
STEPS = 5


def _weights():
    from homan_amd import synth
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    lw["lw_scale_obj"] = 10.0
    return lw


def _models(mano, seeds, frames=4, size=64, obj="cube"):
    from homan_amd import synth
    from homan_amd.jointopt import build_model
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    out = []
    for s in seeds:
        c = synth.make_clip(seed=s, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                            hand_verts_fn=hand_fn)
        out.append(build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                               objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True,
                               optimize_object_scale=True, image_size=size, mano_model=mano, rend_size=size,
                               sync_metrics=False))
    return out


def _rank_main(rank, world, port, out_dir, num_clips, shared, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    if backend == "nccl":                          # one GPU per rank, RCCL over xGMI (a multi-GPU node only)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)                   # every rank on the one GPU
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from homan_amd import dist as hdist
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    mine = hdist.shard_clips(num_clips, rank, world)
    models = _models(mano, [40 + i for i in mine])
    evo = hdist.optimize_clip_shard(models, _weights(), STEPS, shared_scale=shared)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"scale_{rank}.npy"),
            np.asarray([m.int_scales_object.detach().cpu().numpy()[0] for m in models], np.float32))
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), np.asarray([e["loss"] for e in evo], np.float64).reshape(len(models), STEPS))
    np.save(os.path.join(out_dir, f"rot_{rank}.npy"),
            np.asarray([m.rotations_object.detach().cpu().numpy() for m in models], np.float32).reshape(len(models), 4 * 6))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, clips, shared, tmp_path, tag, backend="gloo"):
    import torch.multiprocessing as mp
    out = tmp_path / tag
    out.mkdir()
    port = 23000 + (os.getpid() % 4000) + 7 * world + clips + (500 if backend == "nccl" else 0)
    mp.spawn(_rank_main, args=(world, port, str(out), clips, shared, backend), nprocs=world, join=True)
    cat = lambda name: np.concatenate([np.load(out / f"{name}_{r}.npy") for r in range(world)])
    return cat("scale"), cat("loss"), cat("rot")


def test_two_ranks_fused_shared_scale_equal_the_one_process_batch(mano_model, tmp_path):
    """2 ranks x 1 clip, tied scale: replicas bit-identical across the ranks, and everything - scale, loss rows, final poses -
    bit-identical to the SAME two clips optimised as one 2-clip batch in one process (a sum of two floats does not depend
    on who adds them)."""
    from homan_amd.jointopt import FusedStepper
    scale, loss, rot = _spawn(2, 2, True, tmp_path, "w2")
    assert scale[0] == scale[1] and abs(float(scale[0]) - 1.0) > 1e-4
    models = _models(mano_model, [40, 41])
    st = FusedStepper(models, _weights(), 1e-2, STEPS, shared_scale=True)
    st.run(STEPS)
    evo = st.loss_evolution(STEPS)
    np.testing.assert_array_equal(np.asarray([m.int_scales_object.detach().cpu().numpy()[0] for m in models]), scale)
    np.testing.assert_array_equal(np.asarray([e["loss"] for e in evo]), loss)
    np.testing.assert_array_equal(np.asarray([m.rotations_object.detach().cpu().numpy() for m in models]).reshape(2, -1), rot)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between ranks); the 1-GPU box runs the gloo twin above")
def test_two_ranks_on_two_gpus_over_rccl(mano_model, tmp_path):
    """The same protocol with RCCL carrying the collective between two GPUs (nccl had only ever run one-rank groups
    here): 2 ranks x 2 clips through optimize_clip_shard(shared_scale=True), one all-reduce of one fp32 per step on the compute
    stream between the two captured halves.  Replicas bit-identical across the ranks, and the same optimisation as the four
    clips as ONE 4-clip batch in one process."""
    from homan_amd.jointopt import FusedStepper
    scale, loss, rot = _spawn(2, 4, True, tmp_path, "nccl_w2c4", backend="nccl")
    assert np.all(scale == scale[0]) and abs(float(scale[0]) - 1.0) > 1e-4 and np.isfinite(loss).all()
    models = _models(mano_model, [40, 41, 42, 43])
    st = FusedStepper(models, _weights(), 1e-2, STEPS, shared_scale=True)
    st.run(STEPS)
    # (the one-process batch sums the four gradients in one block sum, the ranks sum two and two and RCCL adds the halves: the
    #  tied scalar may differ in its last bit - every step within 1e-6 relative - while each rank's replicas are identical)
    np.testing.assert_allclose(np.asarray([m.int_scales_object.detach().cpu().numpy()[0] for m in models]), scale, rtol=1e-5)
    np.testing.assert_allclose(np.asarray([e["loss"] for e in st.loss_evolution(STEPS)]), loss, rtol=1e-4)


def test_ranks_without_collective_equal_single_clips(mano_model, tmp_path):
    """cfg4's structure: 2 ranks, 3 clips -> shards [2, 1], no collective; every clip bit-identical to optimising it alone"""
    from homan_amd.jointopt import FusedStepper
    scale, loss, rot = _spawn(2, 3, False, tmp_path, "w2c3")
    for i in range(3):
        (m,) = _models(mano_model, [40 + i])
        st = FusedStepper(m, _weights(), 1e-2, STEPS)
        st.run(STEPS)
        np.testing.assert_array_equal(np.asarray(st.loss_evolution(STEPS)["loss"]), loss[i])
        np.testing.assert_array_equal(m.rotations_object.detach().cpu().numpy().reshape(-1), rot[i])


def test_uneven_shards_and_an_empty_rank_through_the_fused_loop(mano_model, tmp_path):
    """3 ranks, 2 clips -> [1, 1, 0] (the empty rank issues the same broadcast + all-reduces); 2 ranks, 3 clips -> [2, 1]"""
    for world, clips in ((3, 2), (2, 3)):
        scale, loss, _ = _spawn(world, clips, True, tmp_path, f"w{world}c{clips}")
        assert scale.size == clips and np.all(scale == scale[0]) and abs(float(scale[0]) - 1.0) > 1e-4
        assert np.isfinite(loss).all()


def _one_rank_nccl():
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 1000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return True


def test_cfg5_full_size_workload(mano_model):
    """BASELINE cfg5 as one rank of it runs: 8 clips x 30 frames x 256^2 of the bottle, step-2 losses, ONE object scale
    tied across the clips, the all-reduce issued through RCCL (one-rank group) between the two captured halves.  Replicas stay
    bit-identical, every clip's rows are finite and its objective falls; the first row of every clip is the un-tied forward
    (the tie only enters through the optimiser), i.e. equals the clip's own single-clip step-2 forward bit for bit."""
    import torch.distributed as dist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    created = _one_rank_nccl()
    try:
        lw = dict(synth.STEP2_LOSS_WEIGHTS)
        seeds = list(range(8))
        models = _models(mano_model, seeds, 30, 256, "bottle")
        st = FusedStepper(models, lw, 1e-2, 6, shared_scale=True)
        st.run(6)
        evo = st.loss_evolution(6)
        s = st.model.int_scales_object.detach().cpu().numpy()
        assert (s == s[0]).all() and abs(float(s[0]) - 1.0) > 1e-4
        for e in evo:
            assert np.isfinite(e["loss"]).all() and all(np.isfinite(v).all() for v in e.values())
        (alone,) = _models(mano_model, [3], 30, 256, "bottle")
        sa = FusedStepper(alone, lw, 1e-2, 1)
        sa.run(1)
        ea = sa.loss_evolution(1)
        for k, v in ea.items():
            assert v[0] == evo[3][k][0], k
    finally:
        if created:
            dist.destroy_process_group()


def test_cfg5_sized_tied_scale_follows_the_eager_tied_loop(mano_model):
    """Two cfg5-sized clips (30 x 256^2, bottle, step-2, tied scale): the fused loop vs the plain autograd tied loop of
    homan_amd.dist over the HIP model - first rows equal to rounding, the shared scalar after 5 steps within 2e-4."""
    import torch.distributed as dist
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, parameter_groups
    created = _one_rank_nccl()
    try:
        lw = dict(synth.STEP2_LOSS_WEIGHTS)
        eager = _models(mano_model, [0, 1], 30, 256, "bottle")
        opts = [torch.optim.Adam(parameter_groups(m, 1e-2)) for m in eager]
        hist = hdist.optimize_clips_shared_scale(eager, opts, lw, STEPS)
        fused = _models(mano_model, [0, 1], 30, 256, "bottle")
        st = FusedStepper(fused, lw, 1e-2, STEPS, shared_scale=True)
        st.run(STEPS)
        evo = st.loss_evolution(STEPS)
        s = st.model.int_scales_object.detach().cpu().numpy()
        assert s[0] == s[1]
        np.testing.assert_allclose(s[0], eager[0].int_scales_object.detach().cpu().numpy()[0], rtol=2e-4)
        np.testing.assert_allclose([evo[0]["loss"][0], evo[1]["loss"][0]], hist[0], rtol=1e-5)
        np.testing.assert_allclose([evo[0]["loss"][1], evo[1]["loss"][1]], hist[1], rtol=5e-3)
    finally:
        if created:
            dist.destroy_process_group()


def test_tied_scale_trajectory_matches_the_cpu_oracle_loop(mano_model):
    """Where the tied scale GOES is the objective's doing, not the implementation's: the fused loop's shared scalar, step by
    step, against the CPU oracle driven through the same tied loop (homan_amd.dist.optimize_clips_shared_scale).  (In the
    cfg5 bench run the scalar grows 1 -> 2.65 over 400 steps: with the object's depth free, scale and depth trade off along
    a valley of the silhouette term, and the contact term - mean tanh of the distance from every hand vertex to the NEAREST
    object vertex - falls when the object's surface comes closer to the hand, i.e. when the object grows; the prior's
    weight is 1e-3.  Adam moves a parameter whose gradient keeps its sign by ~lr per step whatever its size.)"""
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper
    from oracle.jointopt import collate_inputs, make_optimizer
    from oracle.model import OracleHOMan
    lw, steps, seeds = dict(synth.STEP2_LOSS_WEIGHTS), 10, [50, 51]
    sil_fn, hand_fn = synth.hip_clip_fns(mano_model)
    clips = [synth.make_clip(seed=s, frames=4, rend_size=64, image_size=64, obj="cube", silhouette_fn=sil_fn,
                             hand_verts_fn=hand_fn) for s in seeds]
    oms = []
    for c in clips:
        kw = collate_inputs(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]), c["objvertices"],
                            c["objfaces"])
        oms.append(OracleHOMan(camintr=c["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                               optimize_object_scale=True, image_size=64, mano_model=mano_model, rend_size=64, **kw))
    opts = [make_optimizer(m, 1e-2) for m in oms]
    fused = _models(mano_model, seeds)
    st = FusedStepper(fused, lw, 1e-2, steps, shared_scale=True)
    track_h, track_o = [], []
    for _ in range(steps):
        st.run(1)
        hdist.optimize_clips_shared_scale(oms, opts, lw, 1, device="cpu")
        track_h.append(float(st.model.int_scales_object.detach().cpu()[0]))
        track_o.append(float(oms[0].int_scales_object.detach()[0]))
    assert abs(track_o[-1] - 1.0) > 2e-3                                   # it moves ...
    np.testing.assert_allclose(track_h[:3], track_o[:3], atol=2e-5)         # ... identically at first ...
    # ... and the same way afterwards: once a boundary sample of these 64^2 renders has flipped on one side (step 3 or 4) the
    # two runs are different draws of the same optimisation; they stay within ONE Adam step (lr = 1e-2) of each other
    np.testing.assert_allclose(track_h, track_o, atol=1e-2)


def test_bench_gpus_n_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run on 127.0.0.1) and
    reports n_gpus = 2; gloo because the box has one GPU (both ranks share cuda:0)."""
    import json
    import subprocess
    env = dict(os.environ, HOMAN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--frames", "4", "--size", "64", "--multi-clip", "0", "--parity-seeds", "0", "--lockstep", "0",
                        "--steady", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["clips_per_s"] > 0
