"""What do the cross-queue edges of the fused iteration cost?  Steady-state microseconds per iteration of the cfg2 fit (and of
cfg1 / one frame) for the shipped launch graph, for variants of its fork / join structure (environment switches of
homan_amd/fused.py) and for two MEASUREMENT-ONLY launch graphs defined in this file as subclasses of FusedStepper: the silhouette
chain ALONE on one queue (no side stream at all, the hand does not move - a floor, not a fit), and both chains without any edge
between them inside the four-iteration graph (not a fit either).  Same process, same box, steppers built one after the other.
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

from homan_amd.loopcommon import HmAdam, parameter_groups  # noqa: E402


def _split_groups(model, lr, object_side):
    obj = {id(model.translations_object), id(model.rotations_object)}
    gs = (dict(g, params=[p for p in g["params"] if (id(p) in obj) == object_side]) for g in parameter_groups(model, lr))
    return [g for g in gs if g["params"]]


class ObjectChainOnly(FusedStepper):
    """MEASUREMENT ONLY: the silhouette chain alone on ONE queue - no side stream, no fork, no join, Adam over the object's pose
    only.  The hand stays where it starts (results are NOT the fit's); the object's chain sees its real workload.  Its iteration
    time is the floor any re-arrangement of the two chains can reach."""

    def _build_optimizers(self, lr):
        self.opt = HmAdam(_split_groups(self.model, lr, True))
        self.side_own_vo = True

    def forward_backward(self, log=False):
        it = self._iteration_namespace(log)
        self._issue_silhouette_chain(it)
        self._issue_object_backward(it)


class NoEdges(FusedStepper):
    """MEASUREMENT ONLY: both chains, no edge between them inside the K-iteration graph (the side stream reads whatever object
    pose it finds: results are NOT the fit's) - what the two chains cost each other by sharing the GPU, without what their fork /
    join edges cost.  The object's Adam runs on the calling stream, the hand's on the side stream."""

    def _build_optimizers(self, lr):
        self.opt = HmAdam(_split_groups(self.model, lr, True))
        self.opt_hand = HmAdam(_split_groups(self.model, lr, False))
        self.side_own_vo = True
        self.vo_b = torch.zeros_like(self.vo)

    def _capture_step(self):
        self.opt.step(zero_grad=False, log=self._adam_log())
        self.opt_hand.step(zero_grad=False)

    def _capture_iterations(self, K):
        it = self._iteration_namespace(False)
        it.use_aux = False
        main = it.main
        self.side.wait_stream(main)
        for _ in range(K):
            self._issue_silhouette_chain(it)
            self._issue_object_backward(it)
            self.opt.step(zero_grad=False)
        with torch.cuda.stream(self.side):
            for _ in range(K):
                self._issue_hand_forward(it)
                self._issue_pair_terms(it)
                self._issue_hand_backward(it)
                self.opt_hand.step(zero_grad=False, log=self._adam_log())
        main.wait_stream(self.side)


VARIANTS = [("shipped", {}, FusedStepper), ("main_only", {}, ObjectChainOnly), ("no_edges", {}, NoEdges),
            ("side_waits_for_setup", {"HOMAN_SIDE_OWN_VO": "0"}, FusedStepper), ("shipped_again", {}, FusedStepper)]
if os.environ.get("CHAIN_SKIP"):          # (A/B of library builds: the shipped graph and the one-queue floor only)
    VARIANTS = VARIANTS[:2]
extra = os.environ.get("CHAIN_VARIANTS")          # "name:K=V,K=V;name2:K=V"
if extra:
    for item in extra.split(";"):
        name, kv = item.split(":")
        VARIANTS.append((name, dict(x.split("=") for x in kv.split(",")), FusedStepper))
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
    for vname, env, cls in VARIANTS:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                optimize_mano=True, image_size=kw["size"], mano_model=mano, rend_size=kw["size"], sync_metrics=False)
            st = cls(model, dict(lw), 1e-2, 2000)
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
