"""Times the fused loop on a batch of C cfg2/cfg3 clips (one launch per kernel over C*30 frames) -> one JSON line.
usage: python tools/bench_clips.py [--clips 8] [--steps 200] [--warmup 20] [--step2] [--sweep-blocks N]"""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--step2", action="store_true")
    ap.add_argument("--depth", action="store_true", help="with the ordinal depth term (BASELINE configs[1] as worded)")
    ap.add_argument("--sweep-blocks", type=int, default=0)
    ap.add_argument("--object-scale", action="store_true", help="optimize_object_scale=True (one free scale per clip)")
    ap.add_argument("--shared-scale", action="store_true", help="ONE scale tied across the clips (BASELINE cfg5)")
    ap.add_argument("--no-graph", action="store_true", help="issue the iteration launch by launch instead of replaying a hipGraph")
    ap.add_argument("--cfg1", action="store_true", help="silhouette + 2-D keypoint terms only (no pair-wise losses on the side stream)")
    ap.add_argument("--lw", action="append", default=[], help="override a loss weight, e.g. --lw lw_inter=0")
    ap.add_argument("--groups", type=int, default=1, help="split the clips into this many clip batches, each with its own hipGraph, "
                    "replayed CONCURRENTLY on streams of their own (kernels of different stages of different groups overlap)")
    ap.add_argument("--mixed", action="store_true", help="a heterogeneous shard through dist.optimize_clip_shard's ShardStepper: the "
                    "clips alternate between four shapes (bottle 30 x 256^2, cube 30 x 256^2, bottle 20 x 256^2, cube 20 x 256^2)")
    ap.add_argument("--stamps", type=int, default=0, help="after the timed region: this many more replays with in-kernel "
                    "timestamps -> durations of raster / lines / sweep inside the graph")
    args = ap.parse_args()
    import torch
    from homan_amd import lib as hlib
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    from homan_amd.mano_assets import synthetic_mano
    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    lw = dict(synth.CFG1_LOSS_WEIGHTS if args.cfg1 else synth.STEP2_LOSS_WEIGHTS if args.step2 else synth.STEP1_LOSS_WEIGHTS)
    if args.depth:
        lw["lw_depth"] = 1.0
    for kv in args.lw:
        k, v = kv.split("=")
        lw[k] = float(v)
    models = []
    for i in range(args.clips):
        obj, frames = (("bottle", args.frames), ("cube", args.frames), ("bottle", 2 * args.frames // 3),
                       ("cube", 2 * args.frames // 3))[i % 4] if args.mixed else ("bottle", args.frames)
        c = synth.make_clip(seed=i, frames=frames, rend_size=args.size, image_size=args.size, obj=obj,
                            silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
        models.append(build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]),
                                  objvertices=c["objvertices"], objfaces=c["objfaces"], camintr=c["camintr"],
                                  optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                                  sync_metrics=False, optimize_object_scale=args.object_scale or args.shared_scale,
                                  ordinal_depth=args.depth))
    if args.sweep_blocks:
        hlib.lib().hm_tune_sweep_blocks(args.sweep_blocks)
    total = args.warmup + args.steps
    if args.mixed:
        from homan_amd.jointopt import ShardStepper
        sh = ShardStepper(models, lw, 1e-2, total)
        sh.run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh.run(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        evo = sh.loss_evolution(total)
        print(json.dumps(dict(clips=args.clips, shape_groups=len(sh.steppers), steps=args.steps, ms_per_round=1e3 * el / args.steps,
                              its_per_s=args.clips * args.steps / el, concurrent=os.environ.get("HOMAN_SHARD_CONCURRENT", "1") != "0",
                              final_loss=[e["loss"][-1] for e in evo])))
        return
    if args.groups > 1:
        per = args.clips // args.groups
        sts = [FusedStepper(models[g * per:(g + 1) * per] if per > 1 else models[g * per], lw, 1e-2, total) for g in range(args.groups)]
        streams = [torch.cuda.Stream() for _ in sts]

        def run(n):
            cur = torch.cuda.current_stream()
            for s_ in streams:
                s_.wait_stream(cur)
            for _ in range(n):
                for st_, s_ in zip(sts, streams):
                    with torch.cuda.stream(s_):
                        st_.graph.replay()
            for s_ in streams:
                cur.wait_stream(s_)
        run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        evo = []
        for st_ in sts:
            e = st_.loss_evolution(total)
            evo += e if isinstance(e, list) else [e]
        print(json.dumps(dict(clips=args.clips, groups=args.groups, steps=args.steps, step2=args.step2, ms_per_round=1e3 * el / args.steps,
                              its_per_s=args.clips * args.steps / el, us_per_clip_iteration=1e6 * el / args.steps / args.clips,
                              first_loss=[e["loss"][0] for e in evo], final_loss=[e["loss"][-1] for e in evo])))
        return
    st = FusedStepper(models if args.clips > 1 else models[0], lw, 1e-2, total, capture=not args.no_graph,
                      shared_scale=args.shared_scale)
    st.run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stamps = None
    if args.stamps:
        import ctypes
        L = hlib.lib()
        sctx = st.model.sil_ctx
        ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
        us3, acc = (ctypes.c_float * 3)(), [0.0, 0.0, 0.0]
        saved = torch.zeros(args.stamps, L.hm_sil_timestamps_bytes(*dims) // 8, dtype=torch.int64, device="cuda")
        for i in range(args.stamps):
            hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "ts")
            st.run(1)
            hlib.check(L.hm_sil_timestamps_save(ws, *dims, saved[i].data_ptr(), hlib.stream()), "save")
        hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "ts")
        for i in range(args.stamps):
            hlib.check(L.hm_sil_timestamps_read(None, *dims, saved[i].data_ptr(), ctypes.cast(us3, ctypes.c_void_p), hlib.stream()), "read")
            for k in range(3):
                acc[k] += us3[k] / args.stamps
        stamps = dict(raster_us=round(acc[0], 1), lines_us=round(acc[1], 1), sweep_us=round(acc[2], 1))
    evo = st.loss_evolution(total)
    evo = evo if isinstance(evo, list) else [evo]
    print(json.dumps(dict(in_graph_us=stamps, clips=args.clips, steps=args.steps, step2=args.step2, depth=args.depth, frames=args.frames, rend_size=args.size,
                          faces=int(models[0].faces_object.shape[1]), graph=not args.no_graph, ms_per_round=1e3 * el / args.steps,
                          its_per_s=args.clips * args.steps / el, us_per_clip_iteration=1e6 * el / args.steps / args.clips,
                          first_loss=[e["loss"][0] for e in evo], final_loss=[e["loss"][-1] for e in evo])))


if __name__ == "__main__":
    main()
