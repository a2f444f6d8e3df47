"""What do the cross-queue edges of the fused iteration cost?  Steady-state microseconds per iteration of the cfg2 fit (and of
cfg1 / one frame) for the shipped launch graph, for variants of its fork / join structure (environment switches of
homan_amd/fused.py) and for the silhouette chain ALONE on one queue (HOMAN_EXP_MAIN_ONLY=1: no side stream at all, the hand does
not move - a floor, not a fit).  Same process, same box, steppers built one after the other.
usage (GPU box): python tools/chain_only.py [cfg2 cfg1 b1]"""
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

VARIANTS = [("shipped", {}), ("main_only", {"HOMAN_EXP_MAIN_ONLY": "1"}),
            ("no_edges", {"HOMAN_EXP_NO_EDGES": "1"}), ("side_waits_for_setup", {"HOMAN_SIDE_OWN_VO": "0"}),
            ("shipped_again", {})]
if os.environ.get("CHAIN_SKIP"):          # (A/B of library builds: the shipped graph and the one-queue floor only)
    VARIANTS = VARIANTS[:2]
extra = os.environ.get("CHAIN_VARIANTS")          # "name:K=V,K=V;name2:K=V"
if extra:
    for item in extra.split(";"):
        name, kv = item.split(":")
        VARIANTS.append((name, dict(x.split("=") for x in kv.split(","))))
CONFIGS = dict(cfg1=(dict(frames=10, size=128, obj="cube"), synth.CFG1_LOSS_WEIGHTS),
               b1=(dict(frames=1, size=256, obj="bottle"), synth.CFG1_LOSS_WEIGHTS),
               cfg2=(dict(frames=30, size=256, obj="bottle"), synth.STEP1_LOSS_WEIGHTS),
               cfg3=(dict(frames=30, size=256, obj="bottle"), synth.STEP2_LOSS_WEIGHTS))
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
out = {}
for cname in (sys.argv[1:] or ["cfg2"]):
    kw, lw = CONFIGS[cname]
    clip = synth.make_clip(seed=0, frames=kw["frames"], rend_size=kw["size"], image_size=kw["size"], obj=kw["obj"],
                           silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    out[cname] = {}
    for vname, env in VARIANTS:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                optimize_mano=True, image_size=kw["size"], mano_model=mano, rend_size=kw["size"], sync_metrics=False)
            st = FusedStepper(model, dict(lw), 1e-2, 2000)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        st.run(400)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.run(1200)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / 1200
        # the three heavy kernels inside the replayed graph (hm_sil_timestamps, as bench.py's roofline stamps)
        import ctypes
        from homan_amd import lib as hlib
        L = hlib.lib()
        sctx = st.model.sil_ctx
        ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
        us3, acc, reps = (ctypes.c_float * 3)(), [0.0, 0.0, 0.0], 24
        for _ in range(reps):
            hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "ts")
            st.run(1)
            hlib.check(L.hm_sil_timestamps_read(ws, *dims, None, ctypes.cast(us3, ctypes.c_void_p), hlib.stream()), "ts read")
            acc = [a + float(u) for a, u in zip(acc, us3)]
        hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "ts")
        out[cname][vname] = dict(us_per_iteration=round(us, 2), its_per_s=round(1e6 / us, 1),
                                 raster_lines_sweep_us=[round(a / reps, 1) for a in acc],
                                 final_loss=float(st.loss_evolution(1600)["loss"][-1]))
        del st, model
        sys.stderr.write(f"{cname} {vname}: {us:.1f} us  raster / lines / sweep {out[cname][vname]['raster_lines_sweep_us']}\n")
print(json.dumps(out))
