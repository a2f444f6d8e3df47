#!/usr/bin/env python
"""bench.py -- optimisation iterations / second of the HOMan joint-optimisation hot path on MI355X.

Workload (BASELINE.json configs[1]): one clip per GPU, 30 frames, 256x256 silhouette raster, MANO hand + ~3000-face
bottle, full step-1 loss set, Adam step included.  A "step" = one optimisation iteration (forward + backward +
Adam + loss logging) of one clip, replayed from a hipGraph.  N GPUs = N independent clips (weak scaling, no
data-path collective).  Prints ONE JSON line on rank 0, which also carries
  roofline          the dominant kernel timed with HIP events INSIDE the optimisation loop (launch by launch, next to the
                    other streams' work), algorithmic bytes / that time / 8 TB/s; PMC traffic of the steady-state loop
                    from profiles/ (tools/pmc_loop.sh) when it was measured on the same shapes;
  multi_clip        BASELINE cfg4 in miniature: `--multi-clip` clips per GPU as ONE clip batch (one launch per kernel over
                    all clips), with the whole-iteration roofline fraction of that run;
  final_loss_parity HIP vs the CPU oracle from identical inputs: cfg2 over the steps the CPU baseline leg runs anyway,
                    and BASELINE cfg1 (10 frames 128^2, cube, sil + v2d, 100 steps) in full over several seeds;
  cpu_baseline      the oracle loop timed on this host (bounded sample).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per optimisation iteration per clip, SURVEY.md 8(d) (B=30, S=256, F=3000, V=1502)
def algorithmic_bytes(B, S, F, V, step2=False):
    c_mano = 778 * 3 * 146 * 4 + 2 * 778 * 16 * 4
    obj_transform = 4 * B * V * 12
    mano = 2 * c_mano + 2 * B * 778 * 12
    raster = B * (5 * F * 36 + 2 * (2 * S) ** 2 * 4 + 7 * S * S * 4) + B * V * 12
    v2d = 2 * B * 778 * 20 + B * 778 * 12
    smooth = 2 * B * (V + 778) * 12
    inter = B * (V + 778) * 12
    adam = 28 * 79 * B
    total = obj_transform + mano + raster + v2d + smooth + inter + adam
    if step2:
        total += 2 * B * 32 ** 3 * 4 + B * (778 + V) * 12 + (1552 + F) * 12
        total += 2 * (8 * (778 + V) * 4 * B + B * (778 + V) * 12) + B * 778 * 12
        total += 3 * B * (778 + V) * 12 + 2 * B * 778 * 4
    return dict(total=total, raster=raster)


def kernel_bytes(B, S, F, noaa=False):
    """Algorithmic HBM traffic of ONE launch of the heavy silhouette kernels (every input read once, every output
    written once).
      k_raster_fwd: packed (B,F,3,3) faces + 8-byte boxes + super-region bin lists read; (2S)^2 int32 index map, pooled
                    silhouettes, dimg, alpha bit-plane and the four sweep bit-planes written; keep/ref read.
      k_bwd_lines : four 1-bit planes + dimg read, per-line records {64 mask bits, sources before them} written
                    (16 B per 64 samples); its work-list blocks read faces + boxes + owned flags (46 B per face).
                    (The per-line source arrays and the face records of the work list are data dependent - sources
                    ~0.3 MB, ~45 % of the faces are active - and not counted.)
      k_bwd_sweep : face records of the work list (64 B + 4 B first item, bound: every face), index map, per-line records
                    read; per-face corner gradients (6 doubles: exact sums) written.  (Source arrays as above.)
    noaa (the pose initialisation's fused loop: rendering without anti-aliasing, per-sample masked L2 against ONE binary mask,
    hm_sil_fwd mask_shared = 3 + hm_sil_bwd mode 5): no per-sample image leaves the raster - the index map, the pooled
    silhouette and the bit-planes are its outputs - and the line expansion reads no gradient (it is -1 / +1 by plane)."""
    is_ = 2 * S
    is2 = is_ ** 2
    if noaa:
        return {"k_raster_fwd": B * (F * (36 + 8 + 5) + is2 * 4 + S * S * 4 + 5 * is2 // 8),
                "k_bwd_sweep": B * (F * (64 + 4) + is2 * 4 + is2 + F * 48),
                "k_bwd_lines": B * (4 * is2 // 8 + is2 + F * (36 + 8 + 2))}
    return {"k_raster_fwd": B * (F * (36 + 8 + 5) + is2 * 4 + 4 * S * S * 4 + 5 * is2 // 8),
            "k_bwd_sweep": B * (F * (64 + 4) + is2 * 4 + is2 + F * 48),
            "k_bwd_lines": B * (4 * is2 // 8 + S * S * 4 + is2 + F * (36 + 8 + 2))}


def cpu_baseline(clip, lw, mano, budget_s=20.0, rend_size=256, image_size=256, ordinal_depth=False):
    """Reference CPU path = oracle (CPU restatement) loop on this host, bounded sample, in the two modes of SURVEY 8(d):
    with the per-step `.item()` logging of reference jointopt.py:184-190 (API-faithful; `value`) and without it
    (`value_logging_off`: the same iterations with no host read-back of the losses)."""
    import torch
    from oracle.jointopt import collate_inputs, make_optimizer
    from oracle.model import OracleHOMan
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    torch.set_num_threads(threads)
    kw = collate_inputs(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        clip["objvertices"], clip["objfaces"])
    model = OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                        image_size=image_size, mano_model=mano, rend_size=rend_size, ordinal_depth=ordinal_depth, **kw)
    opt = make_optimizer(model, 1e-2)

    evo = []

    def step(logging=True):
        opt.zero_grad()
        ld, md = model(loss_weights=lw)
        tot = sum(ld[k] * lw[k.replace("loss", "lw")] for k in ld)
        if logging:
            row = {k: v.item() for k, v in ld.items()}
            row["loss"] = tot.item()
            evo.append(row)
        tot.backward()
        opt.step()

    step()                      # warm-up (and step 0 of the trajectory compared in final_loss_parity)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > 0.7 * budget_s or n >= 50:
            break
    m, t1 = 0, time.perf_counter()
    while True:
        step(logging=False)
        m += 1
        el2 = time.perf_counter() - t1
        if el2 > 0.3 * budget_s or m >= 20:
            break
    return dict(value=n / el, unit="it/s", cores=threads, kind="port", value_logging_off=m / el2,
                sample=f"{n} iterations of the same clip ({el:.1f} s) after 1 warm-up, oracle loop with the reference's per-step "
                       f".item() logging; then {m} more iterations ({el2:.1f} s) with the logging off"), evo


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


def bench_shared_scale(args, rank, world, backend, mano, sil_fn, hand_fn):
    """BASELINE cfg5: `--multi-clip` clips per GPU (default 8) with step-2 losses as ONE clip batch per rank, the object
    scale ONE scalar tied across all clips of all ranks: per step one 4-byte all-reduce (sum) of its gradient on the
    compute stream between the two captured halves of the iteration, identical Adam update everywhere."""
    import torch
    import torch.distributed as dist
    from homan_amd import dist as hdist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, build_model
    if not dist.is_initialized():          # N=1: a process group of one rank, so that the RCCL call is really issued
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend, rank=0, world_size=1,
                                **({"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}))
    C = max(1, args.multi_clip)
    lw = dict(synth.STEP2_LOSS_WEIGHTS)
    models = []
    for i in range(C):
        clip = synth.make_clip(seed=100 * rank + i, frames=args.frames, rend_size=args.size, image_size=args.size,
                               obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
        models.append(build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                  objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                                  optimize_mano=True, optimize_object_scale=True, image_size=args.size, mano_model=mano,
                                  rend_size=args.size, sync_metrics=False))
    total = args.warmup + args.steps
    st = FusedStepper(models, lw, 1e-2, total, shared_scale=True)
    st.run(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = hdist.max_over_ranks(time.perf_counter() - t0, device="cuda" if backend == "nccl" else "cpu")
    evo = st.loss_evolution(total)
    evo = evo if isinstance(evo, list) else [evo]
    scale = st.model.int_scales_object.detach().reshape(-1)
    wire = scale[:1].contiguous() if backend == "nccl" else scale[:1].cpu()       # (gloo moves host memory)
    gathered = [torch.zeros_like(wire) for _ in range(world)]
    dist.all_gather(gathered, wire)
    same = bool((scale == scale[0]).all()) and all(torch.equal(g, gathered[0]) for g in gathered)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the same tied-scale semantics as a plain autograd loop over the CPU oracle (homan_amd.dist.optimize_clips_shared_scale,
        # the loop the gloo tests drive), on a bounded sample: two of the clips, a few iterations
        from oracle.jointopt import collate_inputs, make_optimizer
        from oracle.model import OracleHOMan
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
        torch.set_num_threads(threads)
        oms = []
        for i in range(min(C, 2)):
            clip = synth.make_clip(seed=100 * rank + i, frames=args.frames, rend_size=args.size, image_size=args.size,
                                   obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
            kw = collate_inputs(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                clip["objvertices"], clip["objfaces"])
            oms.append(OracleHOMan(camintr=clip["camintr"], class_name="default", int_scale_init=1, optimize_mano=True,
                                   optimize_object_scale=True, image_size=args.size, mano_model=mano, rend_size=args.size, **kw))
        opts = [make_optimizer(m, 1e-2) for m in oms]
        hdist.optimize_clips_shared_scale(oms, opts, lw, 1, device="cpu")          # warm-up
        tc = time.perf_counter()
        hdist.optimize_clips_shared_scale(oms, opts, lw, 3, device="cpu")
        ec = time.perf_counter() - tc
        cpu = dict(value=len(oms) * 3 / ec, unit="it/s (sum over clips)", cores=threads, kind="port",
                   sample=f"{len(oms)} of the clips x 3 tied-scale iterations ({ec:.1f} s) after 1 warm-up, oracle autograd loop")
    if rank == 0:
        F, V = int(models[0].faces_object.shape[1]), int(models[0].verts_object_og.shape[1])
        tot = algorithmic_bytes(args.frames, args.size, F, V, True)["total"]
        value = world * C * args.steps / elapsed
        emit({
            "metric": "optimisation iters/sec (30-frame 256^2 clip)", "value": value,
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg5: {C} clips/GPU x {args.frames} frames {args.size}x{args.size} as one clip batch, "
                                   "step-2 losses, ONE object scale tied across all clips (one 4-byte all-reduce per step, "
                                   f"backend {backend}), fused launch sequence replayed from two hipGraphs around the collective",
                       "clips_per_gpu": C, "parallelism": f"{world * C} clips on {world} ranks, shared scalar"},
            "roofline": dict(bound="hbm", unit="GB/s", peak=8000.0 * world, achieved=tot * value / 1e9,
                             frac=tot * value / (8.0e12 * world), kernel="whole iteration", traffic=None,
                             note="SURVEY 8(d) algorithmic bytes per clip-iteration (cfg3 set) x clip-iterations/s"),
            "cpu_baseline": cpu,
            "shared_scale_final": float(scale[0]), "replicas_identical": same,
            "first_loss": [e["loss"][0] for e in evo], "final_loss": [e["loss"][-1] for e in evo]})
    dist.destroy_process_group()


def pose_init_bench(args):
    """Object-pose initialisation (reference homan/pose_optimization.py:219-383): N poses x steps on one mask."""
    import numpy as np
    import torch
    from homan_amd import pose_optimization as po
    from homan_amd import synth
    n, size, steps = args.pose_init, args.size, (50 if args.steps == 400 else args.steps)
    ov, of = synth.bottle_mesh()
    verts, faces = torch.from_numpy(ov), torch.from_numpy(of).long()
    K = np.array([[480.0, 0, 175.0], [0, 480.0, 175.0], [0, 0, 1.0]], np.float32)
    sq = np.array([75.0, 60.0, 200.0, 200.0], np.float32)
    Rgt = torch.tensor(synth._rot_x(1.3) @ synth._rot_y(0.4), dtype=torch.float32)
    tgt_pose = (verts @ Rgt + torch.tensor([0.0, -0.02, 0.6]))[None]
    roi = po.get_K_crop_resize(torch.as_tensor(K)[None], torch.tensor([[sq[0], sq[1], sq[0] + sq[2], sq[1] + sq[2]]]), [size])
    roi[:, :2] /= size
    tgt_model = po.PoseOptimizer(ref_image=np.zeros((size, size), np.float32), vertices=verts, faces=faces,
                                 rotation_init=po.matrix_to_rot6d(torch.eye(3)[None]), translation_init=torch.zeros(1, 1, 3), K=roi)
    from homan_amd import ops
    with torch.no_grad():
        mask = ops.silhouette_render_noaa(tgt_pose.cuda(), tgt_model._K_all, tgt_model._sil_ctx).cpu().numpy()[0]
    ys, xs = np.nonzero(mask > 0)
    bbox = np.array([sq[0] + xs.min() * sq[2] / size, sq[1] + ys.min() * sq[2] / size,
                     (xs.max() - xs.min()) * sq[2] / size, (ys.max() - ys.min()) * sq[2] / size], np.float32)
    torch.manual_seed(0)
    rots = po.compute_random_rotations(n)
    fit = lambda k, mode: po.find_optimal_pose(verts, faces, mask, bbox, sq, (350, 350), K=K, num_iterations=k,
                                               num_initializations=n, rotations_init=rots, rend_size=size, mode=mode)
    # the loops of find_optimal_pose: "eager" = the reference's loop verbatim (torch autograd + Adam, one host sync per step),
    # "graph" = that step captured in a hipGraph, "fused" (the default of find_optimal_pose) = the step as a fixed C-ABI launch
    # sequence without the autograd tape, in a hipGraph.  The fastest is reported, all are listed.
    loops, best, cold = {}, None, None
    for mode in os.environ.get("HOMAN_POSEINIT_LOOPS", "eager,graph,fused").split(","):
        fit(3, mode)                               # warm-up (allocations, lazy init)
        torch.cuda.synchronize()
        if mode == "fused":
            # the fused loop is run by a RESIDENT fitter (po.PoseFitter, one per mesh / candidate count / mask size - what the
            # per-frame fits of find_optimal_poses share): the timed fit below reuses the one the warm-up built.  A fit that
            # builds everything itself (the first frame of a clip) is timed beside it.
            os.environ["HOMAN_POSE_FITTER"] = "0"
            t0 = time.perf_counter()
            fit(steps, mode)
            torch.cuda.synchronize()
            cold = time.perf_counter() - t0
            del os.environ["HOMAN_POSE_FITTER"]
        t0 = time.perf_counter()
        fitted = fit(steps, mode)
        torch.cuda.synchronize()
        loops[mode] = time.perf_counter() - t0
        if best is None or loops[mode] < loops[best]:
            best, model = mode, fitted
    el = loops[best]
    with torch.no_grad():
        _, iou, _ = model()
    # roofline of the dominant kernel of the fused step, measured inside its replayed hipGraph (the kernels stamp the device wall
    # clock, as for the headline): a fit of the same candidates, its last `reps` replays stamped
    reps = 20
    trans0 = po.TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(verts.unsqueeze(0), rots), torch.as_tensor(K)[None]).unsqueeze(1)
    sm = po.PoseOptimizer(ref_image=mask, vertices=verts, faces=faces, rotation_init=po.matrix_to_rot6d(rots),
                          translation_init=trans0, num_initializations=n, K=roi)
    stamps = po._fused_loop(sm, 1e-2, max(steps - reps, 3), stamp_reps=reps)[3]
    kb = kernel_bytes(n, size // 2, int(faces.shape[0]), noaa=True)
    pmc = {}
    ppath = os.path.join(ROOT, "profiles", "r04_pmc_poseinit.json")
    if os.path.exists(ppath):
        pj = json.load(open(ppath))
        if pj.get("shape") == dict(poses=n, size=size, faces=int(faces.shape[0])):
            pmc = pj.get("per_launch", {})
    per = {}
    for name, us in stamps.items():
        rec = dict(avg_launch_us=us, algorithmic_bytes=kb[name], achieved_GBps=kb[name] / (us * 1e-6) / 1e9)
        c = pmc.get(name, {})
        if c.get("traffic_bytes"):
            rec["traffic_bytes"] = c["traffic_bytes"]
        if c.get("SQ_INSTS_VALU"):
            rec["valu_wave_instr"] = c["SQ_INSTS_VALU"]
            rec["valu_frac"] = c["SQ_INSTS_VALU"] / (us * 1e-6 * 1024 * 2.4e9 / 4)
        per[name] = rec
    dom = max(per, key=lambda k: per[k]["avg_launch_us"])
    roof = dict(bound="hbm", kernel=dom, achieved=per[dom]["achieved_GBps"], peak=8000.0, unit="GB/s",
                frac=per[dom]["achieved_GBps"] / 8000.0, traffic=per[dom].get("traffic_bytes"),
                avg_launch_us=per[dom]["avg_launch_us"], kernels=per,
                timing=f"device wall clock stored by every workgroup at entry and exit in the last {reps} replays of a "
                       f"{steps}-step fit's hipGraph (hm_sil_timestamps)",
                traffic_source="profiles/r04_pmc_poseinit.json (rocprofv3 --pmc passes, tools/pmc_poseinit.sh)" if per[dom].get("traffic_bytes") else None)
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import poseopt
        torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "1")))
        nc, kc = min(n, 64), 10
        poseopt.find_optimal_pose(verts, faces, mask, bbox, sq, (350, 350), K=K, num_iterations=1, num_initializations=nc,
                                  rotations_init=rots[:nc].cpu(), rend_size=size)
        t1 = time.perf_counter()
        poseopt.find_optimal_pose(verts, faces, mask, bbox, sq, (350, 350), K=K, num_iterations=kc, num_initializations=nc,
                                  rotations_init=rots[:nc].cpu(), rend_size=size)
        ec = time.perf_counter() - t1
        cpu = dict(value=nc * kc / ec, unit="pose-steps/s", cores=int(os.environ.get("OMP_NUM_THREADS", "1")), kind="port",
                   sample=f"{nc} poses x {kc} steps of the same fit ({ec:.1f} s), oracle find_optimal_pose")
    emit({"metric": "object-pose initialisation, pose-steps/sec (N poses x one 256^2 mask)",
                      "value": n * steps / el, "unit": "pose-steps/s", "n_gpus": 1, "steps": steps, "warmup": 3,
                      "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"SURVEY 8f rank 1: find_optimal_pose, {n} poses, lathe bottle (3000 faces), "
                                             f"{size}x{size} mask, no anti-aliasing, Adam step in the timed region, "
                                             f"loop = {best} (eager / graph: torch autograd + Adam over the HIP rasteriser; "
                                             f"fused: C-ABI launch sequence in a hipGraph, run by the resident fitter "
                                             f"find_optimal_pose keeps per mesh - the state of every fit of a clip but its "
                                             f"first; `cold_fit` = a fit that builds everything itself)",
                                 "poses": n, "rend_size": size,
                                 "pose_steps_per_s_by_loop": {k: n * steps / v for k, v in loops.items()},
                                 "cold_fit": (dict(seconds_per_fit=cold, pose_steps_per_s=n * steps / cold) if cold else None)},
                      "best_iou": float(iou.max()), "seconds_per_fit": el, "roofline": roof, "cpu_baseline": cpu})


_REAL_STDOUT = None


def _quiet_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes its version banner to fd 1 when
    a process group comes up), so everything but the final line is routed to stderr."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(line) + "\n").encode())


def _launch_ranks(n):
    """Re-executes this command under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port);
    rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=_REAL_STDOUT if _REAL_STDOUT is not None else None)
    if r.returncode:
        raise SystemExit(r.returncode)


def main():
    _quiet_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--step2", action="store_true", help="cfg3: add lw_collision / lw_contact")
    ap.add_argument("--loop", choices=["fused", "graph"], default="fused",
                    help="fused: fixed C-ABI launch sequence (no autograd tape); graph: HOMan.forward + autograd")
    ap.add_argument("--multi-clip", type=int, default=8,
                    help="after the headline run, also time this many clips per GPU optimised as ONE clip batch (one launch "
                         "per kernel over all clips; BASELINE cfg4 has 8 clips per GPU); 0 = skip")
    ap.add_argument("--parity-seeds", type=int, default=5,
                    help="final_loss_parity: number of cfg1 clips (10 frames 128^2, 100 steps) run on both the HIP loop and "
                         "the CPU oracle; 0 = skip")
    ap.add_argument("--shared-scale", action="store_true",
                    help="BASELINE cfg5 instead of the headline: --multi-clip clips per GPU as one clip batch, step-2 losses, "
                         "ONE object scale tied across all clips of all ranks (one 4-byte all-reduce per step inside the "
                         "fused loop)")
    ap.add_argument("--pose-init", type=int, default=0, metavar="N",
                    help="SURVEY 8f rank 1 instead of the headline: one find_optimal_pose fit = N candidate poses of the "
                         "bottle against one 256x256 instance mask, --steps Adam steps (reference default 50); prints "
                         "its own JSON line (pose-steps/sec) with a bounded CPU-oracle baseline")
    ap.add_argument("--depth", action="store_true",
                    help="cfg2 as BASELINE.json words it (sil/kp/depth/smooth): the ordinal depth term of reference "
                         "homan.py:384-419 switched on (lw_depth=1, HOMan(ordinal_depth=True); the reference's own call site "
                         "raises, see DESIGN.md row a19)")
    ap.add_argument("--steady", type=int, default=2000,
                    help="steady_state leg after the headline: the same fit continued to iteration >= 400, then this many "
                         "timed iterations (BASELINE cfg2 is a 400-step fit; a short --steps/--warmup headline times the first, "
                         "heavier iterations); 0 = skip")
    ap.add_argument("--lockstep", type=int, default=24,
                    help="final_loss_parity.lockstep: this many steps of the fused loop re-evaluated by the CPU oracle at the "
                         "HIP parameters (teacher-forced), plus the free-running comparison; 0 = skip")
    ap.add_argument("--freerun", type=int, default=100,
                    help="final_loss_parity.free_run: this many FREE-running steps of the headline clip on the HIP loop and on the "
                         "CPU oracle's reproducible loop - every parameter bit-equal after every step, losses within 1e-4, "
                         "final vertices identical (profiles/ holds 400-step runs of cfg2 and cfg3; with --step2 at most 40 steps, "
                         "the CPU side runs 0.5 it/s); 0 = skip")
    ap.add_argument("--e2e-clips", type=int, default=16,
                    help="end_to_end: this many clips of the headline shape fitted 400 steps each through resident steppers "
                         "(ClipFitter), wall clock from the input dicts to the results on the host; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    if args.pose_init:
        assert args.gpus == 1, "--pose-init is a one-GPU line"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the line the driver
        # uses for N > 1) and hand their single JSON line through.  The children see WORLD_SIZE and take the branch below.
        return _launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, (f"--gpus {args.gpus} but WORLD_SIZE={world}: the rank count comes from the launcher; "
                                f"a mismatch would report {world} GPU(s) as {args.gpus}")
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
    if args.pose_init:
        return pose_init_bench(args)
    import torch
    import torch.distributed as dist
    from homan_amd import synth
    from homan_amd.jointopt import FusedStepper, GraphStepper, build_model
    from homan_amd.mano_assets import synthetic_mano

    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    backend = os.environ.get("HOMAN_BENCH_BACKEND", "nccl")     # "gloo" only to exercise the N>1 path on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            assert world <= ndev, f"{world} ranks but {ndev} GPUs visible"
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    if args.shared_scale:
        return bench_shared_scale(args, rank, world, backend, mano, sil_fn, hand_fn)
    clip = synth.make_clip(seed=rank, frames=args.frames, rend_size=args.size, image_size=args.size, obj="bottle",
                           silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    lw = dict(synth.STEP2_LOSS_WEIGHTS if args.step2 else synth.STEP1_LOSS_WEIGHTS)
    if args.depth:
        lw["lw_depth"] = 1.0
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                        optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                        sync_metrics=False, ordinal_depth=args.depth)
    fused = args.loop == "fused"
    steady_warm = max(0, 400 - (args.warmup + args.steps)) if args.steady > 0 and fused else 0
    stamp_reps = max(10, min(50, args.steps))
    total_steps = args.warmup + args.steps + (2 * stamp_reps + steady_warm + args.steady if fused else 0)
    stepper = (FusedStepper if fused else GraphStepper)(model, lw, 1e-2, total_steps)
    stepper.run(args.warmup)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        stepper.run(n)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    elapsed = timed(args.steps)
    B, S = args.frames, args.size
    F, V = clip["objfaces"].shape[1], clip["objvertices"].shape[1]

    # --- roofline of the dominant kernel, measured INSIDE the replayed graph: ROCm allows no timing events in a captured graph,
    #     so the three heavy kernels stamp the device wall clock themselves (hm_sil_timestamps: every workgroup stores
    #     s_memrealtime at entry and exit into a slot pair of its own; one scalar load per workgroup when switched off, as in
    #     the timed region).  `reps` more replays of THE SAME graph, back to back (arming and saving are stream-ordered 48-byte device
    #     operations, no host synchronisation): same launches, same overlap with the hand-side stream, the state the
    #     timed region left behind.  The rocprofv3 --kernel-trace averages of this command (profiles/) are the cross-check.
    def stamp_roofline(its_per_s, reps):
        import ctypes
        from homan_amd import lib as hlib
        L = hlib.lib()
        us3 = (ctypes.c_float * 3)()
        acc = [0.0, 0.0, 0.0]
        sctx = stepper.model.sil_ctx
        ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
        saved = torch.zeros(reps, L.hm_sil_timestamps_bytes(*dims) // 8, dtype=torch.int64, device="cuda")
        for i in range(reps):          # back to back like the timed region: arm (async memsets), replay, save (48-byte device copy)
            hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "hm_sil_timestamps")
            stepper.run(1)
            hlib.check(L.hm_sil_timestamps_save(ws, *dims, saved[i].data_ptr(), hlib.stream()), "hm_sil_timestamps_save")
        hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "hm_sil_timestamps")
        for i in range(reps):
            hlib.check(L.hm_sil_timestamps_read(None, *dims, saved[i].data_ptr(), ctypes.cast(us3, ctypes.c_void_p), hlib.stream()),
                       "hm_sil_timestamps_read")
            for k in range(3):
                acc[k] += us3[k] * 1e-3           # ms
        torch.cuda.synchronize()
        kb = kernel_bytes(B, S, F)
        pmc, psrc = {}, None
        for cand in ("r04_pmc_loop.json", "r04_pmc_loop_cfg3.json", "r03_pmc_loop.json", "r02_pmc_loop.json"):
            ppath = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(ppath):
                pj = json.load(open(ppath))
                if pj.get("shape") == dict(frames=B, rend_size=S, faces=int(F), step2=bool(args.step2)) and not args.depth:
                    pmc, psrc = pj.get("per_launch", {}), cand      # measured on the same shapes, same steady-state loop
                    break
        per = {}
        for i, name in enumerate(("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep")):
            sec = acc[i] / reps * 1e-3
            rec = dict(avg_launch_us=sec * 1e6, algorithmic_bytes=kb[name], achieved_GBps=kb[name] / sec / 1e9)
            c = pmc.get(name, {})
            if c:
                rec["traffic_bytes"] = c.get("traffic_bytes")
                if c.get("SQ_INSTS_VALU"):
                    # VALU issue roof.  A wave64 VALU instruction runs on a SIMD16 as four passes of 16 lanes: 4 cycles of the
                    # SIMD's VALU per instruction, which is what the counters show (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU, in
                    # quad-cycles per instruction: ~1.0) -> 1024 SIMDs x 2.4 GHz / 4.  The micro-architecture guide's
                    # "wave scheduling" section quotes 2 cycles (dual-issue / packed rate): `valu_frac_2cyc` is the same
                    # count against THAT peak, i.e. half of valu_frac.  DESIGN.md section 5 reconciles the two.
                    rec["valu_wave_instr"] = c["SQ_INSTS_VALU"]
                    rec["valu_frac"] = c["SQ_INSTS_VALU"] / (sec * 1024 * 2.4e9 / 4)
                    rec["valu_frac_2cyc"] = c["SQ_INSTS_VALU"] / (sec * 1024 * 2.4e9 / 2)
                    if c.get("SQ_ACTIVE_INST_VALU"):
                        rec["quad_cycles_per_valu_instr"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
            per[name] = rec
        dom = max(per, key=lambda k: per[k]["avg_launch_us"])
        tot = algorithmic_bytes(B, S, F, V, args.step2)["total"]
        return dict(bound="hbm", kernel=dom, achieved=per[dom]["achieved_GBps"], peak=8000.0, unit="GB/s",
                    frac=per[dom]["achieved_GBps"] / 8000.0, traffic=per[dom].get("traffic_bytes"),
                    avg_launch_us=per[dom]["avg_launch_us"], valu_frac=per[dom].get("valu_frac"),
                    timing=f"device wall clock stored by every workgroup at entry and exit (earliest start to latest end) in "
                           f"{reps} more replays of the timed hipGraph (hm_sil_timestamps); same launches, same overlap",
                    traffic_source=(f"profiles/{psrc} (rocprofv3 --pmc passes over the steady-state loop, tools/pmc_loop.sh)"
                                    if per[dom].get("traffic_bytes") else None),
                    kernels=per,
                    whole_iteration=dict(algorithmic_bytes=tot, achieved_GBps=tot * its_per_s / 1e9,
                                         frac=tot * its_per_s / 8.0e12))

    roof = steady = None
    if fused:
        r = stamp_roofline(args.steps / elapsed, stamp_reps) if rank == 0 else stepper.run(stamp_reps)
        roof = r if rank == 0 else None
        if args.steady > 0:
            # the steady state of the same fit (BASELINE cfg2 is a 400-step fit: iterations 5-25, which short driver flags
            # time, are its heaviest - the band where render and target disagree is still wide, the sweeps see 5-10 M pairs
            # instead of 1.4 M): continue to iteration >= 400, then time `--steady` more
            stepper.run(steady_warm)
            first = args.warmup + args.steps + stamp_reps + steady_warm
            el_s = timed(args.steady)
            sroof = stamp_roofline(args.steady / el_s, stamp_reps) if rank == 0 else stepper.run(stamp_reps)
            if rank == 0:
                steady = dict(value=world * args.steady / el_s, unit="it/s", ms_per_step=1e3 * el_s / args.steady,
                              steps=args.steady, first_timed_iteration=first, seconds=el_s, roofline=sroof,
                              note="same process, same fit, same hipGraph as the headline; max over ranks")

    evo = stepper.loss_evolution(args.warmup + args.steps)

    multi = None
    if args.multi_clip > 1 and args.loop == "fused":
        # BASELINE config 4 in miniature: `multi_clip` independent clips per GPU as ONE clip batch (homan_amd.clipbatch):
        # every kernel is launched once per iteration over all the clips (per-clip normalisers, Adam state, log rows), the
        # whole batched iteration is one hipGraph; every rank its own set, no collective; aggregate = all clips of all
        # ranks / the slowest rank's time.  Bit-identical to optimising the clips one by one (tests/test_clip_batch_gpu.py).
        C, msteps = args.multi_clip, min(args.steps, 200)
        models = []
        for i in range(C):
            ci = synth.make_clip(seed=1000 + 100 * rank + i, frames=args.frames, rend_size=args.size, image_size=args.size,
                                 obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
            models.append(build_model(copy.deepcopy(ci["person_parameters"]), copy.deepcopy(ci["object_parameters"]),
                                      objvertices=ci["objvertices"], objfaces=ci["objfaces"], camintr=ci["camintr"],
                                      optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                                      sync_metrics=False))
        bst = FusedStepper(models, lw, 1e-2, msteps + 10)
        bst.run(10)
        barrier()
        t1 = time.perf_counter()
        bst.run(msteps)
        barrier()
        el = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        mval = world * C * msteps / el
        mtot = algorithmic_bytes(B, S, F, V, args.step2)["total"]
        multi = dict(clips=world * C, clips_per_gpu=C, steps_per_clip=msteps, value=mval,
                     unit="it/s (sum over clips)", ms_per_round=1e3 * el / msteps,
                     vs_single_clip=mval / (world * args.steps / elapsed),
                     roofline=dict(bound="hbm", unit="GB/s", peak=8000.0 * world,
                                   achieved=mtot * mval / 1e9, frac=mtot * mval / (8.0e12 * world),
                                   note="whole iteration: SURVEY 8(d) algorithmic bytes per clip-iteration x clip-iterations/s"),
                     note="one clip batch per GPU: ONE launch per kernel over all clips, one hipGraph per iteration; "
                          "max over ranks")
        del bst, models

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, evo_cpu = cpu_baseline(clip, lw, mano, args.cpu_budget, rend_size=S, image_size=S, ordinal_depth=args.depth)
        parity = dict(cfg2_first_steps=trajectory_parity(evo, evo_cpu),
                      lockstep=(lockstep_parity(mano, step2=args.step2, steps=args.lockstep, frames=B, size=S, clip=clip, lw=lw,
                                                free_run=not args.depth, ordinal_depth=args.depth)
                                if args.lockstep > 0 else None),
                      cfg1=cfg1_parity(mano, seeds=list(range(args.parity_seeds))) if args.parity_seeds > 0 else None,
                      free_run=(free_run_parity(mano, step2=args.step2,
                                                steps=min(args.freerun, 40) if (args.step2 or args.depth) else args.freerun,
                                                frames=B, size=S, clip=clip, lw=lw, ordinal_depth=args.depth)
                                if args.freerun > 0 else None),
                      bar="north_star: 1e-4 relative on losses, 1e-3 mm on final vertices.  cfg1 / free_run compare FREE-running "
                          "trajectories: the object's gradient chain sums in an order-independent way and the hand's chain and "
                          "the step-2 pair terms run in one stated order on both sides (DESIGN.md 2), so EVERY parameter is "
                          "bit-equal after every step (profiles/r04_freerun_cfg{2,3}_400.json: 400 steps); cfg2_first_steps is "
                          "the headline run against the cpu_baseline leg's plain oracle loop (torch Adam, autograd), which "
                          "separates once a sample flips")

    e2e = None
    if rank == 0 and world == 1 and fused and args.e2e_clips > 0 and not args.depth:
        del stepper
        e2e = end_to_end_clips(mano, lw, clips=args.e2e_clips, clips_per_batch=max(1, min(args.multi_clip or 1, args.e2e_clips // 2)),
                               steps=400, frames=B, size=S)
    if rank == 0:
        value = world * args.steps / elapsed
        line = {
            "metric": "optimisation iters/sec (30-frame 256^2 clip)", "value": value, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("cfg3" if args.step2 else "cfg2") +
                       f": 1 clip/GPU x {B} frames {S}x{S}, synthetic MANO hand + lathe bottle ({F} faces, {V} verts), "
                       + ("step-2" if args.step2 else "step-1") + " loss set" + (" + ordinal depth term (lw_depth=1)" if args.depth else "")
                       + ", Adam step + loss logging in the timed region",
                       "frames": B, "rend_size": S, "faces": int(F), "clips_per_gpu": 1,
                       "loop": ("fused C-ABI launch sequence" if args.loop == "fused" else "HOMan.forward + autograd") + ", forward+backward+Adam+logging replayed from a hipGraph", "parallelism": f"{world} independent clips"},
            # BASELINE's second reading of the metric: clips/s (a clip = one 400-step fit) absolute and as the fraction of the
            # N x 8 TB/s HBM roof the SURVEY 8(d) byte model of those iterations amounts to
            "clips_per_s": value / 400.0,
            "clips_per_s_hbm_frac": algorithmic_bytes(B, S, F, V, args.step2)["total"] * value / (8.0e12 * world),
            "final_loss": evo["loss"][-1], "first_loss": evo["loss"][0],
            "roofline": roof, "steady_state": steady, "cpu_baseline": cpu, "multi_clip": multi,
            "final_loss_parity": parity, "end_to_end": e2e,
            "clips_per_s_end_to_end": e2e["clips_per_s_end_to_end"] if e2e else None,
        }
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
