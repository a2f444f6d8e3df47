"""Timing of kernel variants at a FIXED state of the fit (lr = 0: the parameters never move, so builds whose results are wrong -
the ceiling experiments of EXPERIMENTS.md - see exactly the workload of the shipped build).  Two states of the cfg2 clip:
its start (iteration 0: the heaviest) and a converged one (saved by the shipped library after 400 steps).
usage (GPU box):  python tools/ab_state.py save gpurun_out/conv.pt        # shipped library
                  HOMAN_AMD_LIB=variants/lib_x.so python tools/ab_state.py time gpurun_out/conv.pt [clips]"""
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


def model_of(seed):
    clip = synth.make_clip(seed=seed, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    return build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                       objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True,
                       image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)


def rate(models, n):
    st = FusedStepper(models if len(models) > 1 else models[0], dict(synth.STEP1_LOSS_WEIGHTS), 0.0, n + 120)
    st.run(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(n)
    torch.cuda.synchronize()
    return len(models) * n / (time.perf_counter() - t0)


what, path = sys.argv[1], sys.argv[2]
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if what == "save":
    states = []
    for c in range(8):
        m = model_of(c)
        FusedStepper(m, dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, 400).run(400)
        torch.cuda.synchronize()
        states.append({k: v.detach().cpu() for k, v in m.state_dict().items()})
    torch.save(states, path)
else:
    states = torch.load(path)
    out = {"lib": os.environ.get("HOMAN_AMD_LIB", "shipped"), "clips": C}
    start = [model_of(c) for c in range(C)]
    out["start_its"] = round(rate(start, 600 if C == 1 else 150))
    conv = [model_of(c) for c in range(C)]
    for m, s in zip(conv, states):
        m.load_state_dict({k: v.to(next(m.parameters()).device) for k, v in s.items()})
    out["converged_its"] = round(rate(conv, 1500 if C == 1 else 300))
    print(json.dumps(out))
