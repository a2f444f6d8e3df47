"""Standalone launch durations of the three heavy silhouette kernels (raster, sweep, lines) on a batch of C cfg2 clips
after a few optimisation iterations (steady state, persistent outputs): HIP events around back-to-back launches
(hm_bench_sil_kernels).  A/B tool for kernel work; the in-loop numbers come from bench.py.
usage: python tools/bench_sil_kernels.py [--clips 8] [--iters 20] [--reps 20]"""
import argparse
import copy
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--near", type=int, default=-1, help="force the near-winding hint (0/1); -1: as calibrated")
    ap.add_argument("--chain", action="store_true", help="also time the three kernels inside the setup -> raster -> lines -> "
                    "sweeps SEQUENCE of an iteration (in-kernel timestamps), with nothing on any other stream: their cost "
                    "with the data flow of the loop but without its hand-side stream")
    args = ap.parse_args()
    import torch
    from homan_amd import lib as hlib
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    models = []
    for i in range(args.clips):
        c = synth.make_clip(seed=i, frames=args.frames, rend_size=args.size, image_size=args.size, obj="bottle",
                            silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
        models.append(build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                  objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"],
                                  optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                                  sync_metrics=False))
    st = FusedStepper(models if args.clips > 1 else models[0], dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, args.iters + 1)
    st.run(args.iters)
    torch.cuda.synchronize()
    m, sctx = st.model, st.model.sil_ctx
    B, V, F, S = m.B, sctx.V, sctx.F, sctx.S
    verts = st.vo.clone()
    pooled = torch.empty(B, S, S, device="cuda")
    out2 = torch.empty(2, device="cuda")
    gv = torch.empty(B, V, 3, device="cuda")
    one = torch.ones(1, device="cuda")
    ms = torch.zeros(3)
    ksum = m.keep_sum.sum().reshape(1)
    if args.near >= 0:
        hlib.check(hlib.lib().hm_sil_hint_near_winding(hlib.ptr(sctx.workspace), args.near, hlib.stream()), "hint")
    hlib.check(hlib.lib().hm_bench_sil_kernels(
        hlib.ptr(verts), hlib.ptr(sctx.faces), hlib.ptr(m.camintr_rois_object), B, V, F, S, hlib.ptr(m.keep_mask_object),
        hlib.ptr(m.ref_mask_object), hlib.ptr(ksum), hlib.ptr(pooled), hlib.ptr(out2), hlib.ptr(sctx.work_order),
        hlib.ptr(sctx.adj_off), hlib.ptr(sctx.adj_items), None, hlib.ptr(one), hlib.ptr(gv), hlib.ptr(sctx.workspace),
        args.reps, ms.data_ptr(), hlib.stream()), "hm_bench_sil_kernels")
    chain = None
    if args.chain:
        import ctypes
        L = hlib.lib()
        ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
        us3, acc = (ctypes.c_float * 3)(), [0.0, 0.0, 0.0]
        saved = torch.zeros(args.reps, L.hm_sil_timestamps_bytes(*dims) // 8, dtype=torch.int64, device="cuda")
        for i in range(args.reps):
            hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "ts")
            st.sil_chain_only()
            hlib.check(L.hm_sil_timestamps_save(ws, *dims, saved[i].data_ptr(), hlib.stream()), "save")
        hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "ts")
        for i in range(args.reps):
            hlib.check(L.hm_sil_timestamps_read(None, *dims, saved[i].data_ptr(), ctypes.cast(us3, ctypes.c_void_p), hlib.stream()), "read")
            for k in range(3):
                acc[k] += us3[k] / args.reps
        chain = dict(raster_us=acc[0], lines_us=acc[1], sweep_us=acc[2])
    print(json.dumps(dict(chain=chain, clips=args.clips, frames=B, near_calibrated=getattr(sctx, "near_winding", None), near_forced=args.near, raster_us=1e3 * ms[0].item(), sweep_us=1e3 * ms[1].item(),
                          lines_us=1e3 * ms[2].item(), per_clip_us=[1e3 * ms[i].item() / args.clips for i in range(3)])))


if __name__ == "__main__":
    main()
