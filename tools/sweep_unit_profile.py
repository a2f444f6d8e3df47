"""Per-unit profile of k_bwd_sweep (debug build: tools/ab_build.sh uprof -DSWEEP_UNIT_PROFILE;
HOMAN_AMD_LIB=scratch/lib_uprof.so python tools/sweep_unit_profile.py): wall time, face passes, stage-2 trips, queued items
and pair rounds of every unit at iterations 12 and 200 of a cfg2 fit, and a least-squares cost model of a unit.  GPU box."""
import os
import sys, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from homan_amd import lib as _lib, synth
from homan_amd.jointopt import FusedStepper, build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
model = build_model(clip["person_parameters"], clip["object_parameters"], objvertices=clip["objvertices"], objfaces=clip["objfaces"],
                    camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
st = FusedStepper(model, dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, 300)
L = _lib.lib(); L.hm_debug_unit_profile.argtypes = [ctypes.c_void_p]
buf = np.zeros((65536, 4), dtype=np.int32)
done = 0
for at in (12, 200):
    st.run(at - done); done = at
    L.hm_debug_unit_profile(buf.ctypes.data)
    st.run(1); done += 1
    L.hm_debug_unit_profile(buf.ctypes.data)
    a = buf[buf[:, 0] > 0]
    us = a[:, 0] / 100.0; passes = a[:, 1]; trips = a[:, 2] & 0xff; q = a[:, 2] >> 8; rounds = a[:, 3]
    print(f"iter {at}: units {len(a)} us mean {us.mean():.1f} p50 {np.median(us):.1f} p90 {np.percentile(us,90):.1f} p99 {np.percentile(us,99):.1f} max {us.max():.1f}")
    print("  passes hist", np.bincount(passes)[:8], " trips hist", np.bincount(trips)[:8], " rounds p50/p90/p99/max", np.median(rounds), np.percentile(rounds,90), np.percentile(rounds,99), rounds.max())
    for lo, hi in ((0, 8), (8, 12), (12, 16), (16, 20), (20, 30), (30, 1000)):
        m = (us >= lo) & (us < hi)
        if m.sum(): print(f"  units {lo}-{hi} us: n {m.sum()} passes {passes[m].mean():.2f} trips {trips[m].mean():.2f} queued {q[m].mean():.0f} rounds {rounds[m].mean():.2f}")
    # linear fit
    X = np.stack([np.ones_like(us), passes, trips, rounds, q], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(X, us, rcond=None)
    print("  fit us = %.2f + %.2f*passes + %.2f*trips + %.2f*rounds + %.4f*queued" % tuple(coef))
