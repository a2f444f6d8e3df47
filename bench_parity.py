"""Parity and end-to-end legs of the benchmark (opt-in: `python bench.py --parity`, tools/, tests/).

HIP fused loop vs the CPU oracle: cfg1 over several seeds, free-running fits, teacher-forced lock-step, the headline run's first
steps, and the end-to-end ClipFitter walk.  These are the measurements `tests/test_parity_gpu.py` / `tests/test_lockstep_gpu.py`
assert on; bench.py only runs them when asked, and writes their (long) records to a side file, never to its JSON line.
Imports `oracle` - test infrastructure, never on the product path.
"""
import copy
import os
import time

def trajectory_parity(evo_hip, evo_cpu, tol=1e-4):
    """HIP vs oracle loss_evolution from identical inputs: per-loss relative difference per step, the first step at which
    any loss differs by more than `tol` (BASELINE north_star: 1e-4 relative), the differences at the last common step."""
    n = min(len(evo_cpu), len(evo_hip["loss"]))
    keys = [k for k in evo_cpu[0] if k in evo_hip]
    rel = {k: [abs(evo_hip[k][i] - evo_cpu[i][k]) / max(abs(evo_cpu[i][k]), 1e-12) for i in range(n)] for k in keys}
    worst = [max(rel[k][i] for k in keys) for i in range(n)]
    first = next((i for i, w in enumerate(worst) if w > tol), None)
    return dict(steps_compared=n, tol=tol, first_step_over_tol=first, max_rel_diff_step0=worst[0],
                rel_diff_last_step={k: rel[k][n - 1] for k in keys}, max_rel_diff_per_step=worst)


def cfg1_parity(mano, seeds, steps=100, frames=10, size=128):
    """BASELINE cfg1 (the configuration the reference CPU path is defined on): 1 clip, 10 frames 128x128, MANO right hand
    + 1 rigid cube, silhouette + 2-D keypoint losses only, 100 Adam steps.  HIP fused loop vs CPU oracle loop (its reproducible
    form, see free_run_parity) from identical inputs, per seed: final weighted loss of both, relative difference, first step over 1e-4, max final-vertex
    difference (mm); plus the CPU oracle's rate on this configuration."""
    import numpy as np
    import torch
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    from oracle.jointopt import optimize_hand_object as oracle_opt
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    lw = dict(synth.CFG1_LOSS_WEIGHTS)
    rows, cpu_s, gpu_s, control = [], 0.0, 0.0, None
    for seed in seeds:
        clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj="cube", silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
        common = dict(objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                      optimize_mano=True, image_size=size, mano_model=mano, rend_size=size)
        model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                            sync_metrics=False, **common)
        st = FusedStepper(model, lw, 1e-2, steps)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        st.run(steps)
        torch.cuda.synchronize()
        gpu_s += time.perf_counter() - tg
        evo_h = st.loss_evolution(steps)
        t0 = time.perf_counter()
        om, evo_c, _ = oracle_opt(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                  loss_weights=lw, num_iterations=steps, lr=1e-2, reproducible=True, **common)
        cpu_s += time.perf_counter() - t0
        obj_equal = all(np.array_equal(getattr(model, k).detach().cpu().numpy().ravel(), getattr(om, k).detach().numpy().ravel())
                        for k in ("rotations_object", "translations_object"))
        cpu_params = dict(om.named_parameters())
        all_equal = all(np.array_equal(p.detach().cpu().numpy().ravel(), cpu_params[k].detach().numpy().ravel())
                        for k, p in model.named_parameters() if k in cpu_params)
        rel = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(evo_h["loss"], evo_c["loss"])]
        with torch.no_grad():
            dvo = (model.get_verts_object()[0].cpu() - om.get_verts_object()[0]).abs().max().item()
            dvh = (model.get_verts_hand()[0].cpu() - om.get_verts_hand()[0]).abs().max().item()
        if control is None:
            # control experiment: the CPU oracle against ITSELF from inputs that differ by 1e-7 m in one object translation -
            # how far apart two runs of the same implementation end up says how much of the HIP-vs-CPU distance is the
            # algorithm's own sensitivity (piecewise-constant silhouette loss, Adam's normalised steps)
            op2 = copy.deepcopy(clip["object_parameters"])
            op2[0]["translations"] = op2[0]["translations"] + 1e-7
            om2, evo_p, _ = oracle_opt(copy.deepcopy(clip["person_parameters"]), op2, loss_weights=lw, num_iterations=steps,
                                       lr=1e-2, reproducible=True, **common)
            with torch.no_grad():
                control = dict(seed=seed, perturbation_m=1e-7,
                               final_vertex_diff_mm=dict(
                                   object=1e3 * (om2.get_verts_object()[0] - om.get_verts_object()[0]).abs().max().item(),
                                   hand=1e3 * (om2.get_verts_hand()[0] - om.get_verts_hand()[0]).abs().max().item()),
                               rel_diff_final_loss=abs(evo_p["loss"][-1] - evo_c["loss"][-1]) / abs(evo_c["loss"][-1]),
                               first_step_over_tol=next((i for i, (a, b) in enumerate(zip(evo_p["loss"], evo_c["loss"]))
                                                         if abs(a - b) / max(abs(b), 1e-12) > 1e-4), None))
        rows.append(dict(seed=seed, first_loss=evo_c["loss"][0], final_loss_hip=evo_h["loss"][-1],
                         final_loss_cpu=evo_c["loss"][-1], rel_diff_final=rel[-1], rel_diff_step0=rel[0],
                         first_step_over_tol=next((i for i, r in enumerate(rel) if r > 1e-4), None),
                         max_rel_diff_any_step=max(rel), object_params_bit_equal=bool(obj_equal),
                         all_params_bit_equal=bool(all_equal),
                         final_vertex_diff_mm=dict(object=1e3 * dvo, hand=1e3 * dvh)))
    fh, fc = np.array([r["final_loss_hip"] for r in rows]), np.array([r["final_loss_cpu"] for r in rows])
    return dict(config="cfg1: 1 clip, 10 frames 128x128, cube, lw_sil_obj=1 lw_v2d_hand=50, %d Adam steps; HIP fused loop vs the "
                       "CPU oracle's reproducible loop (oracle.jointopt.reproducible_step: the reference loop with the object's "
                       "gradient chain - order-independent sums -, the hand's - one stated order - and Adam written out)" % steps,
                bars=dict(loss_rel=1e-4, vertex_mm=1e-3),
                all_within_bars=all(r["first_step_over_tol"] is None and r["final_vertex_diff_mm"]["object"] < 1e-3
                                    and r["final_vertex_diff_mm"]["hand"] < 1e-3 for r in rows),
                seeds=rows, final_loss_mean=dict(hip=float(fh.mean()), cpu=float(fc.mean())),
                final_loss_std=dict(hip=float(fh.std()), cpu=float(fc.std())),
                max_rel_diff_final=float(max(r["rel_diff_final"] for r in rows)),
                cpu_vs_cpu_control=control,
                cpu_its_per_s=len(seeds) * steps / cpu_s, hip_its_per_s=len(seeds) * steps / gpu_s,
                cores=int(os.environ.get("OMP_NUM_THREADS", "1")))


def free_run_parity(mano, step2=False, steps=100, frames=10, size=128, obj="cube", seed=0, lr=1e-2, lw=None, clip=None,
                    tol=1e-4, stages=True, ordinal_depth=False):
    """BASELINE's end-state bar, free-running: the HIP fused loop and the CPU oracle loop optimise the same clip from identical
    inputs for `steps` iterations, nobody teacher-forced (reference loop: homan/jointopt.py:158-192).

    The oracle runs its REPRODUCIBLE form (oracle.jointopt.reproducible_step): the object's gradient chain (order-independent
    sums), the hand's (one stated order, step-1 loss sets: oracle/handchain.py) and Adam written out - same mathematics as
    autograd + torch.optim.Adam (tests/test_objchain.py), a defined rounding.  The HIP kernels form the same sums (include/homan_amd.h, ORDER-INDEPENDENT SUMS), so on the step-1 loss sets -
    where the object's chain does not depend on the hand (homan/homan.py:482-490) - `rotations_object` /
    `translations_object` must be BIT-EQUAL after every step; reported per step, with the first differing step (None = never),
    the final vertex distances in mm and the relative loss differences (bars: 1e-3 mm, 1e-4)."""
    import numpy as np
    import torch
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    from oracle import objchain
    from oracle.jointopt import collate_inputs, make_optimizer, reproducible_step
    from oracle.model import OracleHOMan
    if clip is None:
        sil_fn, hand_fn = synth.hip_clip_fns(mano)
        clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
    if lw is None:
        lw = dict(synth.STEP2_LOSS_WEIGHTS if step2 else synth.STEP1_LOSS_WEIGHTS)
    if ordinal_depth:
        lw = dict(lw, lw_depth=1.0)
    common = dict(objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True,
                  image_size=size, mano_model=mano, rend_size=size, ordinal_depth=ordinal_depth)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        sync_metrics=False, **common)
    st = FusedStepper(model, lw, lr, steps)
    kw = collate_inputs(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        clip["objvertices"], clip["objfaces"])
    om = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                     image_size=size, mano_model=mano, rend_size=size, ordinal_depth=ordinal_depth, **kw)
    opt = make_optimizer(om, lr, reproducible=True)
    obj_keys = ("rotations_object", "translations_object")
    rows, first_obj_diff, stage_report, first_any_diff = [], None, None, None
    t_cpu = 0.0
    for i in range(steps):
        if stages and first_obj_diff is None:
            before = {k: getattr(om, k).detach().numpy().copy() for k in obj_keys}
        st.run(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ld, md, tot = reproducible_step(om, lw, opt)
        t_cpu += time.perf_counter() - t0
        hip = {k: v[i] for k, v in st.loss_evolution(i + 1).items()}
        cpu = {k: float(v.detach().reshape(-1)[0]) for k, v in ld.items()}
        cpu["loss"] = float(tot.detach().reshape(-1)[0])
        rel = {k: abs(hip[k] - cpu[k]) / max(abs(cpu[k]), 1e-12) for k in cpu if k in hip}
        hp = {k: p.detach().cpu().numpy() for k, p in model.named_parameters()}
        cp = {k: p.detach().numpy().copy() for k, p in om.named_parameters()}
        eq = {k: bool(np.array_equal(hp[k], cp[k].reshape(hp[k].shape))) for k in obj_keys}
        differing = sorted(k for k in hp if k in cp and not np.array_equal(hp[k], cp[k].reshape(hp[k].shape)))
        if first_any_diff is None and differing:
            first_any_diff = dict(step=i, parameters=differing)
        pdiff = {k: float(np.abs(hp[k] - cp[k].reshape(hp[k].shape)).max()) for k in hp if k in cp}
        if first_obj_diff is None and not all(eq.values()):
            first_obj_diff = i
            if stages:
                # where along the chain did step i differ?  The oracle's chain re-evaluated at the parameters BEFORE the step
                # against what the HIP loop left behind (gradients, per-corner sums, vertices, index map)
                for k in obj_keys:
                    getattr(om, k).data.copy_(torch.from_numpy(before[k]))
                g_c, stg = objchain.object_pose_grads(om, lw, return_stages=True)
                sctx = st.model.sil_ctx
                parts_h = sctx.parts().cpu().numpy()
                stage_report = dict(
                    step=i,
                    verts_equal=bool(np.array_equal(st.vo.cpu().numpy(), stg["verts"])),
                    idx_map_differing=int((sctx.idx_map().cpu().numpy() != stg["idx"]).sum()),
                    parts_equal=bool(np.array_equal(parts_h, stg["parts"])) if "parts" in stg else None,
                    parts_max_abs_diff=float(np.abs(parts_h - stg["parts"]).max()) if "parts" in stg else None,
                    parts_differing=int((parts_h != stg["parts"]).sum()) if "parts" in stg else None,
                    parts_max_abs=float(np.abs(stg["parts"]).max()) if "parts" in stg else None,
                    grads_equal={k: bool(np.array_equal(getattr(model, k).grad.cpu().numpy().reshape(g_c[k].shape), g_c[k]))
                                 for k in obj_keys},
                    grads_max_rel={k: float(np.abs(getattr(model, k).grad.cpu().numpy().reshape(g_c[k].shape) - g_c[k]).max()
                                            / max(np.abs(g_c[k]).max(), 1e-30)) for k in obj_keys})
                for k in obj_keys:      # (put the oracle back on its own trajectory)
                    getattr(om, k).data.copy_(torch.from_numpy(cp[k]))
        rows.append(dict(step=i, object_bit_equal=all(eq.values()), max_rel_loss=max(rel.values()), rel=dict(rel),
                         values_cpu={k: cpu[k] for k in rel},
                         worst_loss=max(rel, key=rel.get), max_param_diff=max(pdiff.values()),
                         worst_param=max(pdiff, key=pdiff.get)))
    with torch.no_grad():
        dvo = 1e3 * (model.get_verts_object()[0].cpu() - om.get_verts_object()[0]).abs().max().item()
        dvh = 1e3 * (model.get_verts_hand()[0].cpu() - om.get_verts_hand()[0]).abs().max().item()
    frames, obj = len(clip["object_parameters"]), f"{clip['objfaces'].shape[1]} faces"
    return dict(config=f"{frames} frames {size}x{size}, {obj}, " + ("step-2" if step2 else "step-1 / custom") +
                f" loss set{' + ordinal depth term' if ordinal_depth else ''}, {steps} free-running steps: HIP fused loop vs the CPU oracle's reproducible loop",
                steps=steps, tol=tol, first_step_object_params_differ=first_obj_diff,
                object_params_bit_equal_all_steps=first_obj_diff is None,
                first_step_any_param_differs=first_any_diff, all_params_bit_equal_all_steps=first_any_diff is None,
                first_step_over_tol=next((r["step"] for r in rows if r["max_rel_loss"] > tol), None),
                max_rel_loss=max(r["max_rel_loss"] for r in rows), worst_loss=max(rows, key=lambda r: r["max_rel_loss"])["worst_loss"],
                final_rel_loss=rows[-1]["max_rel_loss"], final_vertex_diff_mm=dict(object=dvo, hand=dvh),
                final_max_param_diff=rows[-1]["max_param_diff"], final_worst_param=rows[-1]["worst_param"],
                stage_report=stage_report, cpu_its_per_s=steps / max(t_cpu, 1e-9),
                first_over_tol_detail=(lambda j: None if j is None else dict(
                    step=j, rel_at_step=rows[j]["rel"], rel_step_before=rows[j - 1]["rel"] if j else None,
                    values_cpu_at_step=rows[j]["values_cpu"], values_cpu_step_before=rows[j - 1]["values_cpu"] if j else None,
                    worst_param_at_step=rows[j]["worst_param"], max_param_diff_before=rows[j - 1]["max_param_diff"] if j else None))(
                    next((r["step"] for r in rows if r["max_rel_loss"] > tol), None)),
                cores=int(os.environ.get("OMP_NUM_THREADS", "1")),
                per_step=[{k: r[k] for k in ("step", "object_bit_equal", "max_rel_loss", "max_param_diff")} for r in rows][:: max(1, steps // 25)])


def end_to_end_clips(mano, lw, clips=16, clips_per_batch=8, steps=400, frames=30, size=256, seed0=2000):
    """BASELINE cfg4's clips/s, END TO END: `clips` cfg2-shaped clips fitted `steps` iterations each through resident steppers
    (homan_amd.jointopt.ClipFitter, the sample loop of reference fit_vid_dataset.py:190-379), timed from the per-frame input
    dicts on the host to the results (parameters, vertices, loss_evolution) back on the host - model build, workspace
    allocation, calibration and graph capture included for the first batch of a shape, input load + replay + read-back for
    the others.  Generating the synthetic clips (the dataset / detector side) is outside the timed region."""
    import torch
    from homan_amd import synth
    from homan_amd.jointopt import ClipFitter
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    data = [synth.make_clip(seed=seed0 + i, frames=frames, rend_size=size, image_size=size, obj="bottle", silhouette_fn=sil_fn,
                            hand_verts_fn=hand_fn) for i in range(clips)]
    fitter = ClipFitter(lw, num_iterations=steps, optimize_mano=True, image_size=size, mano_model=mano, rend_size=size,
                        clips_per_batch=clips_per_batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fitter.fit(data)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = fitter.timing
    first = clips_per_batch                       # the clips of the batch that built the stepper
    reused_clips = clips - first
    per_reused = (t["load"] + (t["iterations"] + t["read_back"]) * reused_clips / clips) / max(reused_clips, 1)
    its = t["iterations"] / clips
    return dict(clips=clips, clips_per_batch=clips_per_batch, steps_per_clip=steps, seconds=el, clips_per_s_end_to_end=clips / el,
                split_s={k: t[k] for k in ("collate", "build", "load", "iterations", "read_back")},
                steppers_built=t["built"], batches_reused=t["reused"],
                repeated_shape=dict(seconds_per_clip=per_reused, clips_per_s=1.0 / per_reused,
                                    setup_fraction_of_fit=(t["load"] / max(reused_clips, 1)) / its,
                                    note="a clip of a shape already resident: input load + its share of the replays + read-back"),
                final_loss_mean=float(sum(r["loss_evolution"]["loss"][-1] for r in res) / clips),
                what="ClipFitter: one resident stepper (buffers, workspaces, ONE hipGraph) per shape signature; wall clock from "
                     "the input dicts to the results on the host")


def lockstep_parity(mano, step2=False, steps=50, frames=30, size=256, obj="bottle", seed=0, lr=1e-2, free_run=True,
                    clip=None, lw=None, tol=1e-4, ordinal_depth=False):
    """Teacher-forced parity along the HIP trajectory (reference loop: homan/jointopt.py:158-192).

    The fused loop runs `steps` iterations one replay at a time.  BEFORE every step its parameters are loaded into the CPU
    oracle, which evaluates THAT step there: loss_dict (bar 1e-4 relative), parameter gradients (error / largest entry),
    camera-space vertices (mm) and the face-index map of the silhouette raster (samples whose owner differs).  Every step is
    a single-step comparison at identical parameters, so no trajectory can hide in it; the hard rasteriser's chaos only
    enters through what the comparison measures - a flipped sample.
    With `free_run` a second oracle optimises from the same start with torch's Adam (the reference loop): the distance of
    the two FREE trajectories per step (parameters in ulps / absolute, samples that differ, weighted loss) says when they
    separate and the lock-step numbers of the step before say what differed first."""
    import numpy as np
    import torch
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    from oracle import nmr as o_nmr
    from oracle.jointopt import collate_inputs, make_optimizer
    from oracle.model import OracleHOMan
    if clip is None:
        sil_fn, hand_fn = synth.hip_clip_fns(mano)
        clip = synth.make_clip(seed=seed, frames=frames, rend_size=size, image_size=size, obj=obj, silhouette_fn=sil_fn,
                               hand_verts_fn=hand_fn)
    if lw is None:
        lw = dict(synth.STEP2_LOSS_WEIGHTS if step2 else synth.STEP1_LOSS_WEIGHTS)
    if ordinal_depth:           # cfg2 as BASELINE.json words it (sil / kp / depth / smooth): reference homan.py:384-419
        lw = dict(lw, lw_depth=1.0)
    common = dict(objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True,
                  image_size=size, mano_model=mano, rend_size=size)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        sync_metrics=False, ordinal_depth=ordinal_depth, **common)
    st = FusedStepper(model, lw, lr, steps)

    def oracle_model():
        kw = collate_inputs(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                            clip["objvertices"], clip["objfaces"])
        return OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                           image_size=size, mano_model=mano, rend_size=size, ordinal_depth=ordinal_depth, **kw)

    def oracle_depth_idx(om):
        """face-index maps of the two depth renders of the ordinal depth term (object, hand) at the full-image camera"""
        with torch.no_grad():
            r = o_nmr.Renderer(image_size=size, K=om.camintr, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
            out = []
            for v, fc in ((om.get_verts_object()[0], om.faces_object),
                          (om.get_verts_hand()[0], om.faces_hand[0][None].repeat(om.camintr.shape[0], 1, 1))):
                f = r._ndc_faces(v, fc, om.camintr, None, None, None, None)
                out.append(o_nmr._RasterizeAlphaDepth.apply(f, 2 * size, r.near, r.far, r.rasterizer_eps)[2].numpy())
            return out

    def oracle_idx(om):
        with torch.no_grad():
            r = om.losses.renderer
            f = r._ndc_faces(om.get_verts_object()[0], om.faces_object, om.camintr_rois_object, None, None, None, None)
            return o_nmr._RasterizeAlphaDepth.apply(f, 2 * size, r.near, r.far, r.rasterizer_eps)[2].numpy()

    def fwd_bwd(om):
        for p in om.parameters():
            p.grad = None
        ld, md = om(loss_weights=lw)
        tot = sum(ld[k] * lw[k.replace("loss", "lw")] for k in ld)
        tot.sum().backward()
        row = {k: float(v.detach().reshape(-1)[0]) for k, v in ld.items()}
        row.update({k: float(v) for k, v in md.items()})
        row["loss"] = float(tot.detach().reshape(-1)[0])
        return row

    forced = oracle_model()
    free = oracle_model() if free_run else None
    opt = make_optimizer(free, lr) if free_run else None
    sctx = st.model.sil_ctx
    rows, free_rows = [], []
    for i in range(steps):
        params = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
        st.run(1)
        torch.cuda.synchronize()
        hip = st.loss_evolution(i + 1)
        hip = {k: v[i] for k, v in hip.items()}
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        idx_h = sctx.idx_map().cpu().numpy()
        didx_h = [st.dctx[0].idx_map().cpu().numpy(), st.dctx[1].idx_map().cpu().numpy()] if ordinal_depth else None
        vo_h, vh_h = st.vo.cpu().numpy(), st.vh.cpu().numpy()
        forced.load_state_dict(params, strict=False)
        cpu = fwd_bwd(forced)
        with torch.no_grad():
            vo_c, vh_c = forced.get_verts_object()[0].numpy(), forced.get_verts_hand()[0].numpy()
        rel = {k: abs(hip[k] - cpu[k]) / max(abs(cpu[k]), 1e-12) for k in cpu if k in hip and k.startswith("loss")}
        # logged metrics (not part of the objective).  handobj_maxdist: the reference forms |a|^2 + |b|^2 - 2ab in fp32
        # (libyana batch_pairwise_dist, losses.py:227), whose own rounding is ~4e-6 m at a 6 mm gap; the kernel differences
        # the coordinates first -> compared in metres
        met = {k: abs(hip[k] - cpu[k]) / max(abs(cpu[k]), 1e-12) for k in cpu if k in hip and not k.startswith("loss")}
        maxdist_abs = abs(hip["handobj_maxdist"] - cpu["handobj_maxdist"]) if "handobj_maxdist" in cpu and "handobj_maxdist" in hip else 0.0
        gerr = {}
        for k, p in forced.named_parameters():
            if p.grad is None or k not in grads:
                continue
            ref = p.grad.numpy()
            gerr[k] = float(np.abs(grads[k] - ref).max() / max(np.abs(ref).max(), 1e-30))
        worst_loss = max(rel, key=rel.get)
        worst_grad = max(gerr, key=gerr.get)
        col_given = None
        if "loss_collision" in cpu:
            # the same oracle term evaluated on the HIP loop's HAND vertices (1 ulp from the oracle's: the MANO sums run in
            # another order; the object's vertices are bit-equal): what is left of the difference is the SDF kernels'
            from oracle import model as o_model
            with torch.no_grad():
                cg = float(o_model.compute_collision_loss(torch.from_numpy(vh_h), torch.from_numpy(vo_h), forced.faces_object,
                                                          forced.closed_faces)["loss_collision"])
            col_given = abs(hip["loss_collision"] - cg) / max(abs(cg), 1e-12)
        rows.append(dict(step=i, max_rel_loss=rel[worst_loss], worst_loss=worst_loss, worst_loss_value=cpu[worst_loss],
                         weighted_share=abs(hip[worst_loss] - cpu[worst_loss]) * lw[worst_loss.replace("loss", "lw")] / max(abs(cpu["loss"]), 1e-12)
                         if worst_loss != "loss" else rel[worst_loss], max_grad_err=gerr[worst_grad],
                         worst_grad=worst_grad, flipped_samples=int((idx_h != oracle_idx(forced)).sum()),
                         vert_diff_mm=dict(object=1e3 * float(np.abs(vo_h - vo_c).max()), hand=1e3 * float(np.abs(vh_h - vh_c).max())),
                         vert_equal=dict(object=bool(np.array_equal(vo_h, vo_c)), hand=bool(np.array_equal(vh_h, vh_c))),
                         rel_loss=rel, rel_metric=met, handobj_maxdist_abs_m=maxdist_abs,
                         collision_rel_given_hip_vertices=col_given,
                         flipped_depth_samples=([int((a != b).sum()) for a, b in zip(didx_h, oracle_depth_idx(forced))]
                                                if ordinal_depth else None)))
        if free_run:
            # the free-running reference loop, one step behind the comparison: its parameters BEFORE its step i against the
            # HIP loop's parameters before step i
            fp = {k: p.detach().numpy() for k, p in free.named_parameters()}
            dist = {k: float(np.abs(fp[k] - params[k].numpy()).max()) for k in fp if k in params}
            wk = max(dist, key=dist.get)
            fidx = oracle_idx(free)
            opt.zero_grad()
            frow = fwd_bwd(free)
            opt.step()
            free_rows.append(dict(step=i, max_param_diff=dist[wk], worst_param=wk,
                                  samples_differing=int((idx_h != fidx).sum()),
                                  rel_diff_total=abs(hip["loss"] - frow["loss"]) / max(abs(frow["loss"]), 1e-12),
                                  rel_diff_worst=max(abs(hip[k] - frow[k]) / max(abs(frow[k]), 1e-12)
                                                     for k in frow if k in hip and k.startswith("loss"))))
    out = dict(config=("cfg3" if step2 else "cfg2") + f"-shaped: {frames} frames {size}x{size}, {obj}, "
               + ("step-2" if step2 else "step-1") + " loss set" + (" + ordinal depth term" if ordinal_depth else "") + f", {steps} steps of the fused loop, every step re-evaluated by "
               "the CPU oracle at the HIP parameters", steps=steps, tol=tol,
               max_rel_loss=max(r["max_rel_loss"] for r in rows), max_grad_err=max(r["max_grad_err"] for r in rows),
               flipped_samples=sum(r["flipped_samples"] for r in rows),
               # (object: its vertices are bit-equal, so is its depth render; hand: vertices one ulp apart, a sample may flip)
               flipped_depth_samples=(dict(object=sum(r["flipped_depth_samples"][0] for r in rows),
                                           hand=sum(r["flipped_depth_samples"][1] for r in rows)) if ordinal_depth else None),
               max_vert_diff_mm=dict(object=max(r["vert_diff_mm"]["object"] for r in rows),
                                     hand=max(r["vert_diff_mm"]["hand"] for r in rows)),
               object_vertices_bit_equal=all(r["vert_equal"]["object"] for r in rows),
               hand_vertices_bit_equal=all(r["vert_equal"]["hand"] for r in rows),
               max_rel_metric={k: max(r["rel_metric"].get(k, 0.0) for r in rows) for k in rows[0]["rel_metric"]},
               max_handobj_maxdist_abs_m=max(r["handobj_maxdist_abs_m"] for r in rows),
               max_collision_rel_given_hip_vertices=(max(r["collision_rel_given_hip_vertices"] for r in rows)
                                                     if rows[0]["collision_rel_given_hip_vertices"] is not None else None),
               first_step_over_tol=next((r["step"] for r in rows if r["max_rel_loss"] > tol), None),
               worst_loss_per_key={k: max(r["rel_loss"].get(k, 0.0) for r in rows) for k in rows[0]["rel_loss"]},
               worst_grad_per_step=[(r["worst_grad"], r["max_grad_err"]) for r in rows][:8],
               per_step=[{k: r[k] for k in ("step", "max_rel_loss", "worst_loss", "worst_loss_value", "weighted_share",
                                            "max_grad_err", "worst_grad", "flipped_samples")} for r in rows])
    if free_run:
        sep = next((r["step"] for r in free_rows if r["rel_diff_worst"] > tol), None)
        first_flip = next((r["step"] for r in free_rows if r["samples_differing"] > 0), None)
        out["free_run"] = dict(
            what="HIP fused loop vs the CPU oracle loop (torch Adam), both free-running from identical inputs",
            first_step_over_tol=sep, first_step_with_differing_samples=first_flip,
            max_param_diff_per_step=[r["max_param_diff"] for r in free_rows][:12],
            samples_differing_per_step=[r["samples_differing"] for r in free_rows][:12],
            rel_diff_worst_per_step=[r["rel_diff_worst"] for r in free_rows][:12],
            at_separation=(free_rows[sep] if sep is not None else None),
            before_separation=(dict(lockstep=rows[sep - 1]["max_grad_err"], worst_grad=rows[sep - 1]["worst_grad"],
                                    free=free_rows[sep - 1]) if sep else None),
            final=free_rows[-1])
    return out
