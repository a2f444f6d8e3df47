"""Phases of the MANO backward (k_mano_bwd) on the cfg2 clip, stand-alone (debug build: tools/ab_build.sh mph -DMANO_PHASES;
HOMAN_AMD_LIB=scratch/lib_mph.so python tools/mano_phases.py).  GPU box."""
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
st = FusedStepper(m, dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, 1000, capture=False)
L = hlib.lib(); P = hlib.ptr
L.hm_debug_mano_phases.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 16)()
st.run(5)
mm = st.model; pca, rot, betas, mtr = mm.mano_pca_pose, mm.mano_rot, mm.mano_betas, mm.mano_trans
def call():
    hlib.check(L.hm_mano_bwd(st.mctx.ptrs, P(pca), st.P, P(rot), P(betas), st.B, P(st.G_mesh), P(st.U_pca), 1e-4, P(pca.grad), P(rot.grad), P(betas.grad),
                             P(mtr.grad), P(st.mano_state), P(st.mctx.workspace(st.B)), hlib.stream()), "mano_bwd")
for _ in range(3): call()
L.hm_debug_mano_phases(out)
torch.cuda.synchronize(); t = time.perf_counter()
N = 200
for _ in range(N): call()
torch.cuda.synchronize(); el = (time.perf_counter() - t) / N * 1e6
L.hm_debug_mano_phases(out)
wg, last = max(1, int(out[15])), max(1, int(out[14]))
names = ["state load", "skin + grads", "3 block sums", "dA", "dfeat rows", "ticket"]
print(f"{el:.1f} us per launch back to back; per workgroup (us): " + "  ".join(f"{n}={int(out[k]) / wg / 100.0:.2f}" for k, n in enumerate(names)) + f"  | second half (last workgroup of a frame): {int(out[6]) / last / 100.0:.2f}")
