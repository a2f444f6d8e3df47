"""GPU box: free-running HIP loop vs the CPU oracle's reproducible loop (bench.free_run_parity), a few configurations.
usage: python tools/chain_parity.py [cfg1|cfg2small|cfg2|cfg2depth|cfg3|cfg2ctrl] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
import bench_parity as bench  # noqa: E402
from homan_amd import synth  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mano = synthetic_mano(0)
if which == "cfg1":
    out = bench.free_run_parity(mano, steps=steps, frames=10, size=128, obj="cube", lw=dict(synth.CFG1_LOSS_WEIGHTS))
elif which == "cfg2ctrl":
    # control: the CPU oracle's reproducible loop against ITSELF, cfg2 at full size, from hand translations 1e-7 m apart - how
    # far the HAND's own dynamics (Adam at 10 x lr on the MANO parameters) carry a difference of the size of one ulp
    import copy
    import numpy as np
    import torch
    from homan_amd import synth as sy
    from oracle.jointopt import collate_inputs, make_optimizer, reproducible_step
    from oracle.model import OracleHOMan
    sil_fn, hand_fn = sy.hip_clip_fns(mano)
    clip = sy.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    lw = dict(sy.STEP1_LOSS_WEIGHTS)
    models = []
    for eps in (0.0, 1e-7):
        pp = copy.deepcopy(clip["person_parameters"])
        for p in pp:
            p["translations"] = p["translations"] + eps
        kw = collate_inputs(pp, copy.deepcopy(clip["object_parameters"]), clip["objvertices"], clip["objfaces"])
        om = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True, image_size=256,
                         mano_model=mano, rend_size=256, **kw)
        models.append((om, make_optimizer(om, 1e-2, reproducible=True)))
    rows = []
    for i in range(steps):
        tot = [float(reproducible_step(om, lw, opt)[2].detach().reshape(-1)[0]) for om, opt in models]
        with torch.no_grad():
            dh = 1e3 * (models[0][0].get_verts_hand()[0] - models[1][0].get_verts_hand()[0]).abs().max().item()
            do = 1e3 * (models[0][0].get_verts_object()[0] - models[1][0].get_verts_object()[0]).abs().max().item()
        rows.append(dict(step=i, rel_loss=abs(tot[0] - tot[1]) / abs(tot[0]), hand_mm=dh, object_mm=do))
    out = dict(what="CPU oracle (reproducible loop) vs itself, cfg2 full size, hand translations perturbed by 1e-7 m", steps=steps,
               final=rows[-1], first_step_hand_over_bar=next((r["step"] for r in rows if r["hand_mm"] > 1e-3), None),
               per_step=rows[:: max(1, steps // 25)])
elif which == "cfg2depth":
    out = bench.free_run_parity(mano, steps=steps, frames=30, size=256, obj="bottle", ordinal_depth=True)
elif which == "cfg3":
    out = bench.free_run_parity(mano, step2=True, steps=steps, frames=30, size=256, obj="bottle")
elif which == "cfg2small":
    out = bench.free_run_parity(mano, steps=steps, frames=6, size=64, obj="bottle")
else:
    out = bench.free_run_parity(mano, steps=steps, frames=30, size=256, obj="bottle")
print(json.dumps(out))
