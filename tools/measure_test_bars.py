"""What do the loose trajectory tests actually measure?  (parity housekeeping: bars at measured x 1.5.)
Prints the largest relative deviations of tests/test_model_gpu.py::test_short_trajectory_eager_and_graph and
::test_cfg1_full_fit_follows_the_reference_loop on this GPU.   usage (GPU box): python tools/measure_test_bars.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import test_model_gpu as T  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

mano = synthetic_mano(0)
out = {}
from homan_amd.jointopt import FusedStepper, GraphStepper, parameter_groups  # noqa: E402
for name in ["ref_step1_cube_b4_s64", "ref_step2_cube_b4_s64", "ref_step2_twohands_cube_b4_s64", "ref_step1_lefthand_cube_b4_s64"]:
    for mode in ("eager", "graph"):
        rec, model, weights, meta = T._build_hip(name, mano, sync=(mode == "eager"))
        steps = 6
        if mode == "eager":
            opt = torch.optim.Adam(parameter_groups(model, meta["lr"]))
            evo = []
            for _ in range(steps):
                opt.zero_grad()
                ld, _ = model(loss_weights=weights)
                tot = sum(ld[k] * weights[k.replace("loss", "lw")] for k in ld)
                evo.append(tot.item())
                tot.sum().backward()
                opt.step()
        else:
            st = GraphStepper(model, weights, meta["lr"], steps)
            st.run(steps)
            evo = st.loss_evolution(steps)["loss"]
        ref = rec["evo_loss"][:steps]
        rel = np.abs(np.asarray(evo) - ref) / np.abs(ref)
        out[f"{name}:{mode}"] = dict(first=float(rel[0]), first3=float(rel[:3].max()), all6=float(rel.max()))
rec, model, weights, meta = T._build_hip("ref_cfg1_cube_b10_s128", mano, sync=False)
st = FusedStepper(model, weights, meta["lr"], meta["steps"])
st.run(meta["steps"])
evo = st.loss_evolution(meta["steps"])
for k in ("loss", "loss_sil_obj", "loss_v2d_hand"):
    got, ref = np.asarray(evo[k]), rec["evo_" + k]
    out["cfg1:" + k] = dict(max_rel=float((np.abs(got - ref) / np.abs(ref)).max()),
                            max_abs_over_first=float(np.abs(got - ref).max() / abs(float(ref[0]))),
                            final_rel=float(abs(got[-1] - ref[-1]) / abs(ref[-1])))
print(json.dumps(out, indent=1))
