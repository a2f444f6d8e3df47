"""BASELINE cfg1 (10 frames 128^2, cube, silhouette + 2-D keypoints) through the fused loop: microseconds per iteration in the
steady state - the floor of the iteration (its kernels are nearly empty: what is left is their own latency chains).
usage (GPU box): python tools/cfg1_floor.py [cfg1|b1|cfg2 ...]      (default: cfg1 cfg2)
  b1 = ONE frame of the cfg2 clip (256^2, bottle, silhouette + 2-D keypoints): every kernel of the chain at its own latency floor
tools/ledger.sh runs each configuration under rocprofv3 --kernel-trace for the per-kernel ledger of DESIGN.md section 5."""
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from homan_amd import synth  # noqa: E402
from homan_amd.jointopt import FusedStepper, build_model  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
out = {}
CONFIGS = dict(cfg1=(dict(frames=10, size=128, obj="cube"), synth.CFG1_LOSS_WEIGHTS),
               b1=(dict(frames=1, size=256, obj="bottle"), synth.CFG1_LOSS_WEIGHTS),
               cfg2=(dict(frames=30, size=256, obj="bottle"), synth.STEP1_LOSS_WEIGHTS))
for name in (sys.argv[1:] or ["cfg1", "cfg2"]):
    kw, lw = CONFIGS[name]
    clip = synth.make_clip(seed=0, frames=kw["frames"], rend_size=kw["size"], image_size=kw["size"], obj=kw["obj"],
                           silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True,
                        image_size=kw["size"], mano_model=mano, rend_size=kw["size"], sync_metrics=False)
    st = FusedStepper(model, dict(lw), 1e-2, 2600)
    st.run(400)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(2000)
    torch.cuda.synchronize()
    out[name] = dict(us_per_iteration=1e6 * (time.perf_counter() - t0) / 2000)
    out[name]["its_per_s"] = 1e6 / out[name]["us_per_iteration"]
print(json.dumps(out))
