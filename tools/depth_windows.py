import copy, json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from homan_amd import synth
from homan_amd.jointopt import FusedStepper, build_model
from homan_amd.mano_assets import synthetic_mano
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
lw = dict(synth.STEP1_LOSS_WEIGHTS)
if depth:
    lw["lw_depth"] = 1.0
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), objvertices=clip["objvertices"],
                    objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano,
                    rend_size=256, sync_metrics=False, ordinal_depth=bool(depth))
st = FusedStepper(model, lw, 1e-2, 4000)
out = []
for w in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.run(200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    evo = st.loss_evolution((w + 1) * 200)
    out.append((w * 200, round(200 / dt), round(evo["loss"][-1], 5), round(evo.get("loss_depth", [0])[-1], 6)))
print(out)
