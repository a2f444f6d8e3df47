"""GPU box: what the driver's flags (--steps 20 --warmup 5) time - iterations 5..25 of a fresh fit - against later windows of the
same fit, with and without the device spun up beforehand.  usage: python tools/early_iters.py [spin_ms]"""
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
from homan_amd import synth  # noqa: E402
from homan_amd.jointopt import FusedStepper, build_model  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

spin_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
lw = dict(synth.STEP1_LOSS_WEIGHTS)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), objvertices=clip["objvertices"],
                    objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano,
                    rend_size=256, sync_metrics=False)
st = FusedStepper(model, lw, 1e-2, 2000)
if spin_ms > 0:
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        for _ in range(10):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()
st.run(5)
rows = []
for w in range(12):
    n = 20 if w < 6 else 100
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(n)
    torch.cuda.synchronize()
    rows.append(dict(n=n, its_per_s=n / (time.perf_counter() - t0)))
print(json.dumps(dict(spin_ms=spin_ms, windows=rows)))
