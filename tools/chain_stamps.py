"""Where the one-clip iteration's time goes INSIDE the replayed graph, launch by launch on the silhouette chain (debug build:
tools/ab_build.sh chain -DHM_CHAIN_STAMPS; HOMAN_AMD_LIB=scratch/lib_chain.so python tools/chain_stamps.py [--step2]): first-start
/ last-end device wall clock of the face setup, the three heavy kernels (hm_sil_timestamps), both gradient launches and the
Adam step of ONE replay, next to the period of back-to-back replays.  GPU box."""
import argparse, copy, ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser(); ap.add_argument("--step2", action="store_true"); ap.add_argument("--warm", type=int, default=300)
args = ap.parse_args()
import torch, numpy as np
from homan_amd import lib as hlib, synth
from homan_amd.jointopt import FusedStepper, build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
c = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
m = build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]), objvertices=c["objvertices"],
                objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
st = FusedStepper(m, dict(synth.STEP2_LOSS_WEIGHTS if args.step2 else synth.STEP1_LOSS_WEIGHTS), 1e-2, args.warm + 1000)
L = hlib.lib(); sctx = st.model.sil_ctx
for f in ("hm_debug_chain_raster", "hm_debug_chain_geometry", "hm_debug_chain_adam"):
    getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_int]
ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
n64 = L.hm_sil_timestamps_bytes(*dims) // 8
saved = torch.zeros(n64, dtype=torch.int64, device="cuda")
st.run(args.warm)
torch.cuda.synchronize(); t = time.perf_counter(); st.run(400); torch.cuda.synchronize()
period = (time.perf_counter() - t) / 400 * 1e6
rows = []
for rep in range(5):
    for f in ("hm_debug_chain_raster", "hm_debug_chain_geometry", "hm_debug_chain_adam"): getattr(L, f)(None, 1)
    hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "ts")
    torch.cuda.synchronize()
    st.run(1)
    hlib.check(L.hm_sil_timestamps_save(ws, *dims, saved.data_ptr(), hlib.stream()), "save")
    hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "ts")
    torch.cuda.synchronize()
    o = [(ctypes.c_ulonglong * 8)() for _ in range(3)]
    L.hm_debug_chain_raster(o[0], 0); L.hm_debug_chain_geometry(o[1], 0); L.hm_debug_chain_adam(o[2], 0)
    raw = saved.cpu().numpy().astype(np.int64)
    nr = sctx.B * 32 * 32 // 4
    def se(blk):
        blk = blk.reshape(-1, 2); blk = blk[(blk[:, 0] > 0) & (blk[:, 1] > 0)]
        return int(blk[:, 0].min()), int(blk[:, 1].max())
    ev = dict(setup=(int(o[0][0]), int(o[0][1])), raster=se(raw[:2 * nr]), lines=se(raw[2 * nr:-2 * 4 * 4096]), sweep=se(raw[-2 * 4 * 4096:]),
              grad_hand=(int(o[1][0]), int(o[1][1])), grad_obj=(int(o[1][2]), int(o[1][3])), adam=(int(o[2][0]), int(o[2][1])))
    t0 = ev["setup"][0]
    rows.append({k: ((v[0] - t0) / 100.0, (v[1] - t0) / 100.0) for k, v in ev.items()})
print(f"period of back-to-back replays: {period:.1f} us")
for k in ("setup", "raster", "lines", "sweep", "grad_hand", "grad_obj", "adam"):
    a = np.array([r[k] for r in rows])
    print(f"  {k:10s} start {np.median(a[:, 0]):7.1f}  end {np.median(a[:, 1]):7.1f}  ({np.median(a[:, 1] - a[:, 0]):5.1f} us)")
