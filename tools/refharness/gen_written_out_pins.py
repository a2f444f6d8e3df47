"""Regression pins for the WRITTEN-OUT form of the oracle (oracle.model.REFERENCE_FORM = False): the loss values it produces
on the inputs of every reference-run golden -> tests/golden/pins_written_out.npz.  The reference-run goldens bound this form at
2e-5 (loss_collision: 1e-4, a few trilinear SDF samples that move by ~5e-5 of themselves per ulp of a hand vertex); these
pins hold it to ITSELF at 1e-6, so that a change of its evaluation order shows up as such.
usage: python tools/refharness/gen_written_out_pins.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from homan_amd.mano_assets import synthetic_mano  # noqa: E402
from oracle import model as o_model  # noqa: E402
from tests import util  # noqa: E402

o_model.REFERENCE_FORM = False
mano = synthetic_mano(0)
out = {}
for name in util.golden_names():
    rec, inputs, camintr, weights, meta = util.load_golden(name)
    m = o_model.OracleHOMan(mano_model=mano, rend_size=meta["image_size"], **util.model_kwargs(inputs, camintr, meta))
    ld, _ = m(loss_weights=weights)
    for k, v in ld.items():
        out[f"{name}/{k}"] = np.float64(v.detach().reshape(-1)[0].item())
    print(name, {k: float(v.detach().reshape(-1)[0]) for k, v in ld.items() if k in ("loss_collision", "loss_contact")})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pins_written_out.npz"), **out)
