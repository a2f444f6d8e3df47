"""world_size-2 gloo test of the multi-GPU path's logic on CPU: clip sharding + the shared-object-scale all-reduce
(BASELINE cfg5).  The per-clip model here is the CPU oracle (the HIP model needs a GPU); the distributed helpers
under test are the product's (homan_amd.dist)."""
import copy
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
STEPS, NUM_CLIPS, LR = 3, 2, 1e-2


def _build(seed, mano):
    from homan_amd import synth
    from oracle.jointopt import collate_inputs, make_optimizer
    from oracle.model import OracleHOMan
    from tests import util
    sil_fn, hand_fn = util.oracle_clip_fns(mano)
    clip = synth.make_clip(seed=seed, frames=4, rend_size=32, image_size=32, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    model = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                        optimize_object_scale=True, image_size=32, mano_model=mano, rend_size=32, **kw)
    return model, make_optimizer(model, LR)


def _weights():
    from homan_amd import synth
    lw = dict(synth.STEP1_LOSS_WEIGHTS)
    lw["lw_scale_obj"] = 10.0
    return lw


def _worker(rank, world, port, out_dir, num_clips=NUM_CLIPS):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from homan_amd import dist as hdist
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    mine = hdist.shard_clips(num_clips, rank, world)
    pairs = [_build(seed, mano) for seed in mine]
    models, opts = [p[0] for p in pairs], [p[1] for p in pairs]
    hist = hdist.optimize_clips_shared_scale(models, opts, _weights(), STEPS)
    scales = np.asarray([m.int_scales_object.detach().numpy()[0] for m in models], np.float32)
    np.save(os.path.join(out_dir, f"scale_{rank}.npy"), scales)
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), np.asarray(hist).reshape(STEPS, len(models)))
    dist.destroy_process_group()


def test_shared_scale_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = np.load(tmp_path / "scale_0.npy")[:1]
    s1 = np.load(tmp_path / "scale_1.npy")[:1]
    np.testing.assert_array_equal(s0, s1)               # replicas of the shared scalar stay bit-identical
    assert abs(float(s0[0]) - 1.0) > 1e-4                # ... and it actually moved

    # single-process reference: both clips in one process, gradients of the scalar summed by hand
    sys.path.insert(0, ROOT)
    from homan_amd import dist as hdist
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    pairs = [_build(seed, mano) for seed in range(NUM_CLIPS)]
    hist = hdist.optimize_clips_shared_scale([p[0] for p in pairs], [p[1] for p in pairs], _weights(), STEPS)
    np.testing.assert_allclose(pairs[0][0].int_scales_object.detach().numpy(), s0, rtol=1e-6)
    h0, h1 = np.load(tmp_path / "hist_0.npy"), np.load(tmp_path / "hist_1.npy")
    np.testing.assert_allclose(np.concatenate([h0, h1], 1), np.asarray(hist), rtol=1e-5)


def test_shard_clips_is_balanced_and_complete():
    sys.path.insert(0, ROOT)
    from homan_amd import dist as hdist
    for n, w in ((64, 8), (9, 8), (2, 3), (0, 2), (7, 3)):
        shards = [hdist.shard_clips(n, r, w) for r in range(w)]
        assert sum(shards, []) == list(range(n))                        # contiguous blocks, every clip exactly once
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert [len(hdist.shard_clips(9, r, 8)) for r in range(8)] == [2, 1, 1, 1, 1, 1, 1, 1]


def test_shared_scale_uneven_shards_and_empty_rank(tmp_path):
    """3 ranks, 2 clips -> shards [1, 1, 0]: the rank without a clip issues the same collectives (one broadcast, one
    all-reduce per step, a zero gradient) and nobody deadlocks; 2 ranks, 3 clips -> shards [2, 1]."""
    for world, clips in ((3, 2), (2, 3)):
        out = tmp_path / f"w{world}"
        out.mkdir()
        port = 31500 + (os.getpid() % 2000) + world
        mp.spawn(_worker, args=(world, port, str(out), clips), nprocs=world, join=True)
        scales = np.concatenate([np.load(out / f"scale_{r}.npy") for r in range(world)])
        assert scales.size == clips and np.all(scales == scales[0]) and abs(float(scales[0]) - 1.0) > 1e-4
        from homan_amd import dist as hdist
        from homan_amd.mano_assets import synthetic_mano
        mano = synthetic_mano(0)
        pairs = [_build(seed, mano) for seed in range(clips)]
        hdist.optimize_clips_shared_scale([p[0] for p in pairs], [p[1] for p in pairs], _weights(), STEPS)
        np.testing.assert_allclose(pairs[0][0].int_scales_object.detach().numpy()[0], scales[0], rtol=1e-6)
