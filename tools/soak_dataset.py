"""Dataset-style soak: many clips fitted one after the other in ONE process, alternating mesh / size; checks finite and
decreasing losses, bounded device memory and a bounded number of live hipGraphs.
  python tools/soak_dataset.py [n] [fitter|fused|graph]
    fitter  (default) the resident steppers of jointopt.ClipFitter: one graph per shape signature however many clips
    fused   optimize_hand_object per clip, step 1 then step 2 with resume (one stepper + graph per fit, the graphs kept alive)
    graph   optimize_hand_object(mode="graph") per clip: HOMan.forward + autograd captured per fit - the captures share one
            memory pool (lib.autograd_pool), so the kept graphs do not pile their activations up"""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from homan_amd import lib, synth
from homan_amd.jointopt import ClipFitter, optimize_hand_object
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
how = sys.argv[2] if len(sys.argv) > 2 else "fitter"
SHAPES = ((10, 128, "cube"), (30, 256, "bottle"), (6, 64, "cube"), (12, 96, "bottle"))
t0 = time.time()
mem, graphs0 = [], len(lib._KEPT_GRAPHS)
fitters = {}
for i in range(n):
    frames, size, obj = SHAPES[i % 4]
    c = synth.make_clip(seed=i, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    kw = dict(objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True, image_size=size,
              mano_model=mano, rend_size=size)
    if how == "fitter":
        f = fitters.setdefault(size, ClipFitter(dict(synth.STEP1_LOSS_WEIGHTS), num_iterations=30, optimize_mano=True,
                                                image_size=size, mano_model=mano, rend_size=size))
        evo1 = f.fit([c])[0]["loss_evolution"]
    else:
        m1, evo1, _ = optimize_hand_object(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                           loss_weights=dict(synth.STEP1_LOSS_WEIGHTS), num_iterations=30,
                                           mode="graph" if how == "graph" else "auto", **kw)
        if how == "fused":
            sd = {k: v for k, v in m1.state_dict().items() if "mano_model" not in k}
            m2, evo2, _ = optimize_hand_object(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                               loss_weights=dict(synth.STEP2_LOSS_WEIGHTS), num_iterations=20, state_dict=sd, **kw)
            assert np.isfinite(evo2["loss"]).all(), i
            del m2
        del m1
    assert np.isfinite(evo1["loss"]).all(), i
    assert evo1["loss"][-1] < evo1["loss"][0], (i, evo1["loss"][0], evo1["loss"][-1])
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_allocated() / 2**20)
    if i % 10 == 9:
        print(f"clip {i + 1}: {time.time() - t0:.0f} s, device memory allocated {mem[-1]:.0f} MiB (after first 4: {mem[3]:.0f}), reserved "
              f"{torch.cuda.memory_reserved() / 2**20:.0f} MiB, live graphs {len(lib._KEPT_GRAPHS) - graphs0}", flush=True)
print("ok", how, n, "clips;", f"{time.time() - t0:.0f} s; memory growth after the first cycle: {mem[-1] - mem[3]:.0f} MiB; "
      f"live graphs {len(lib._KEPT_GRAPHS) - graphs0}")
