#!/usr/bin/env python
"""Run only the silhouette kernels of the cfg2 workload (for rocprofv3 --pmc passes / kernel traces).
Usage: python tools/bench_raster.py [reps] [fwd|bwd|both]"""
import copy
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from homan_amd import lib as hlib  # noqa: E402
from homan_amd import synth  # noqa: E402
from homan_amd.jointopt import build_model  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
what = sys.argv[2] if len(sys.argv) > 2 else "both"
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn,
                       hand_verts_fn=hand_fn)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                    objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                    optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
lw = dict(synth.STEP1_LOSS_WEIGHTS)
for _ in range(reps):
    vo, _ = model.get_verts_object()
    l, m = model.losses.compute_sil_loss_object(vo)
    if what in ("bwd", "both"):
        l["loss_sil_obj"].sum().backward()
torch.cuda.synchronize()
print("done", l["loss_sil_obj"].item())
