"""Where does an active k_raster_fwd workgroup spend its time?  Debug build with per-workgroup phase stamps (no atomics):
    tools/ab_build.sh trace -DRASTER_TRACE ; HOMAN_AMD_LIB=variants/lib_trace.so python tools/raster_trace.py [--frames 30] [--depth]
Prints, for the raster launch of iteration `--warm` of a cfg2-shaped fit (replayed from the captured graph, next to the hand-side
kernels as in the real loop), the phase durations of wave 0 in microseconds - all active workgroups, and the slowest tenth."""
import argparse
import copy
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=400)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--depth", choices=["sil", "obj", "hand"], default=None,
                    help="cfg2 WITH the ordinal depth term: trace the silhouette raster, the object's or the hand's depth render")
    a = ap.parse_args()
    import torch
    from homan_amd import lib as _lib, synth
    from homan_amd.jointopt import FusedStepper, build_model
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    clip = synth.make_clip(seed=0, frames=a.frames, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True,
                        image_size=256, mano_model=mano, rend_size=256, sync_metrics=False, ordinal_depth=a.depth is not None)
    lw = dict(synth.CFG1_LOSS_WEIGHTS if a.frames < 2 else synth.STEP1_LOSS_WEIGHTS)
    if a.depth:
        lw["lw_depth"] = 1.0
    os.environ["HOMAN_GRAPH_ITERS"] = "1"
    st = FusedStepper(model, lw, 1e-2, a.warm + 8)
    L = _lib.lib()
    nwg = a.frames * 256
    buf = (ctypes.c_uint * (9 * nwg))()
    L.hm_debug_raster_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if a.depth:
        F_obj, F_hand = int(clip["objfaces"].shape[1]), 1538
        L.hm_debug_raster_trace_filter(F_hand if a.depth == "hand" else F_obj, 0 if a.depth == "sil" else 1)
    st.run(a.warm)
    L.hm_debug_raster_trace(buf, nwg)          # clear
    st.run(1)
    L.hm_debug_raster_trace(buf, nwg)
    t = np.frombuffer(buf, dtype=np.uint32).reshape(nwg, 9).astype(np.float64)
    ph = t[:, :6] * 0.01                        # 10 ns ticks -> us
    tot = ph.sum(1)
    act = tot > 0
    start = t[:, 6]
    names = ["scan", "records", "near_units", "wait+hz", "far_units", "epilogue"]
    print(f"frames {a.frames} {a.depth or ''}: workgroups {nwg}, active {int(act.sum())}; launch span (first start .. last end of active) "
          f"{(np.max((start + t[:, :6].sum(1))[act]) - np.min(start[act])) * 0.01:.1f} us")
    def show(sel, label):
        p = ph[sel]
        print(f"  {label} ({int(sel.sum())} workgroups): total mean {tot[sel].mean():.2f} p50 {np.median(tot[sel]):.2f} p90 "
              f"{np.percentile(tot[sel], 90):.2f} max {tot[sel].max():.2f} us; candidates near {np.mean(t[sel, 7] % 65536):.0f} far "
              f"{np.mean(t[sel, 7] // 65536):.0f}; units near {np.mean(t[sel, 8] % 65536):.0f} far {np.mean(t[sel, 8] // 65536):.0f}")
        print("     " + "  ".join(f"{n} {p[:, k].mean():.2f}" for k, n in enumerate(names)))
    show(act, "active")
    thr = np.percentile(tot[act], 90)
    show(act & (tot >= thr), "slowest tenth")
    # when do workgroups start?  (a second round of workgroups = the launch's tail)
    s0 = (start[act] - start[act].min()) * 0.01
    print(f"  start offsets of active workgroups: p50 {np.median(s0):.1f} p90 {np.percentile(s0, 90):.1f} max {s0.max():.1f} us")


if __name__ == "__main__":
    main()
