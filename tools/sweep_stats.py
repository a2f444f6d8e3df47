"""Work counters of k_bwd_sweep and phase cycles of k_raster_fwd in the steady state of a cfg2 fit (debug build:
tools/ab_build.sh stats -DSWEEP_STATS -DRASTER_PHASES; HOMAN_AMD_LIB=variants/lib_stats.so python tools/sweep_stats.py).
GPU box."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=100)
    ap.add_argument("--step2", action="store_true")
    ap.add_argument("--frames", type=int, default=30)
    a = ap.parse_args()
    from homan_amd import lib as _lib, synth
    from homan_amd.jointopt import FusedStepper, build_model
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    clip = synth.make_clip(seed=0, frames=a.frames, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    model = build_model(clip["person_parameters"], clip["object_parameters"], objvertices=clip["objvertices"],
                        objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256,
                        mano_model=mano, rend_size=256, sync_metrics=False)
    lw = dict(synth.STEP2_LOSS_WEIGHTS if a.step2 else synth.CFG1_LOSS_WEIGHTS if a.frames < 2 else synth.STEP1_LOSS_WEIGHTS)
    st = FusedStepper(model, lw, 1e-2, a.warm + 8, capture=False)
    L = _lib.lib()
    L.hm_debug_sweep_stats.argtypes = [ctypes.c_void_p]
    L.hm_debug_raster_phases.argtypes = [ctypes.c_void_p]
    rout = (ctypes.c_ulonglong * 24)()
    rnames = ["scan", "near_rec", "near_units", "far_hz_rec", "far_units", "tail", "wg_active", "wg_idle", "units_near",
              "units_far", "-", "-"]
    out = (ctypes.c_ulonglong * 16)()
    names = ["s2_items", "s2_geo", "act0", "act1", "on0", "on1", "pairs", "s2_trips", "s1_items", "s1_geo", "s1_reach",
             "pair_rounds", "s1_own", "s1_a0", "s1_out_empty", "s1_a1"]
    marks = (0, 20, a.warm)
    for step in range(a.warm + 1):
        if step in marks:
            L.hm_debug_sweep_stats(out)          # reset
            L.hm_debug_raster_phases(rout)
        st.run(1)
        if step in marks:
            L.hm_debug_sweep_stats(out)
            print(f"step {step}: " + "  ".join(f"{n}={int(v)}" for n, v in zip(names, out)))
            L.hm_debug_raster_phases(rout)
            tot = sum(int(rout[k]) for k in range(6))
            print(f"   raster wave-0 cycles per active workgroup: " + "  ".join(
                f"{n}={int(v) / max(1, int(rout[6])):.0f}" for n, v in zip(rnames[:6], rout)) +
                f"  (total {tot / max(1, int(rout[6])):.0f});  active {int(rout[6])} idle {int(rout[7])}  units/wg near "
                f"{int(rout[8]) / max(1, int(rout[6])):.0f} far {int(rout[9]) / max(1, int(rout[6])):.0f}  covered pairs "
                f"{int(rout[11])}  wave trips of the covered-sample loop {int(rout[10])} (lane fill {int(rout[11]) / max(1, 64 * int(rout[10])):.2f})")
            print(f"   candidates near {int(rout[12])} far {int(rout[13])}; far candidates with a surviving block {int(rout[20])}, far units "
                  f"surviving the hidden-block test {int(rout[21])} of {int(rout[9])}; units if 4x4 blocks were anchored at the box corner: "
                  f"near {int(rout[14])} (now {int(rout[8])}) far {int(rout[15])}; as 8x2 blocks: near {int(rout[16])} far {int(rout[17])}; box samples "
                  f"near {int(rout[18])} far {int(rout[19])}")


if __name__ == "__main__":
    main()
