"""Dataset-style soak: many clips fitted one after the other in ONE process (one stepper + graph per clip), step 1 then
step 2 with resume, alternating mesh / size; checks finite losses, decreasing loss, bounded device memory."""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from homan_amd import synth
from homan_amd.jointopt import optimize_hand_object
from homan_amd.pose_optimization import find_optimal_pose
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
t0 = time.time()
mem = []
for i in range(n):
    frames, size, obj = ((10, 128, "cube"), (30, 256, "bottle"), (6, 64, "cube"), (12, 96, "bottle"))[i % 4]
    c = synth.make_clip(seed=i, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    kw = dict(objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True, image_size=size,
              mano_model=mano, rend_size=size)
    m1, evo1, _ = optimize_hand_object(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                       loss_weights=dict(synth.STEP1_LOSS_WEIGHTS), num_iterations=30, **kw)
    sd = {k: v for k, v in m1.state_dict().items() if "mano_model" not in k}
    m2, evo2, _ = optimize_hand_object(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                       loss_weights=dict(synth.STEP2_LOSS_WEIGHTS), num_iterations=20, state_dict=sd, **kw)
    assert np.isfinite(evo1["loss"]).all() and np.isfinite(evo2["loss"]).all(), i
    assert evo1["loss"][-1] < evo1["loss"][0], (i, evo1["loss"][0], evo1["loss"][-1])
    del m1, m2
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_allocated() / 2**20)
    if i % 10 == 9:
        print(f"clip {i + 1}: {time.time() - t0:.0f} s, device memory allocated {mem[-1]:.0f} MiB (after first 4: {mem[3]:.0f}), reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)
print("ok", n, "clips x 2 fits;", f"{time.time() - t0:.0f} s; memory growth after the first cycle: {mem[-1] - mem[3]:.0f} MiB")
