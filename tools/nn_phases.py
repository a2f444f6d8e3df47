"""Phases of the metric-only nearest-vertex search (k_nn_min) on the cfg2 clip, stand-alone (debug build:
tools/ab_build.sh nnph -DNN_PHASES; HOMAN_AMD_LIB=scratch/lib_nnph.so python tools/nn_phases.py): wall-clock share of loads /
bounds / first scans / survivor scans / reduction per workgroup and the number of surviving groups.  GPU box."""
import os
import copy, ctypes, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from homan_amd import lib as hlib, synth
from homan_amd.jointopt import FusedStepper, build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
c = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
m = build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]), objvertices=c["objvertices"],
                objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
st = FusedStepper(m, dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, 1000)
L = hlib.lib(); P = hlib.ptr
L.hm_debug_nn_phases.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 10)()
import os
HO = os.environ.get("HO", "1") == "1"
for at in (0, 200):
    st.run(at - (0 if at == 0 else 0))
    mm = st.model
    args = (P(st.vh), P(st.vo), st.B, st.Vh, st.Vo, None, None, st._slot("handobj_maxdist"), P(st.reduce_ws_b.buf), st.clip_len, st.NS,
            P(st.obj_order), P(st.obj_spheres), P(mm.rotations_object), P(mm.translations_object), P(mm.int_scales_object), (P(st.hand_order) if HO else None), (P(st.nn_seed) if os.environ.get('SEED', '1') == '1' else None), hlib.stream())
    for _ in range(3): hlib.check(L.hm_nn_fwd_rigid_clips(*args), "nn")
    L.hm_debug_nn_phases(out)
    torch.cuda.synchronize(); t = time.perf_counter()
    N = 200
    for _ in range(N): hlib.check(L.hm_nn_fwd_rigid_clips(*args), "nn")
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / N * 1e6
    L.hm_debug_nn_phases(out)
    wg = max(1, int(out[6]))
    names = ["loads+spheres", "bounds", "list", "scan", "reduce+ticket", "-"]
    print(f"after {at} its: {el:.1f} us per launch back to back; per workgroup (us): " + "  ".join(f"{n}={int(out[k]) / wg / 100.0:.2f}" for k, n in enumerate(names[:5])) + f"  survivors/wg={int(out[7]) / wg:.2f} of {(st.Vo + 63) // 64}")
