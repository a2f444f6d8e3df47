#!/usr/bin/env python
"""bench.py -- optimisation iterations / second of the HOMan joint-optimisation hot path on MI355X.

Workload (BASELINE.json configs[1]): one clip per GPU, 30 frames, 256x256 silhouette raster, MANO hand + ~3000-face
bottle, full step-1 loss set, Adam step included.  A "step" = one optimisation iteration (forward + backward +
Adam + loss logging) of one clip, replayed from a hipGraph.  N GPUs = N independent clips (weak scaling, no
data-path collective).  Prints ONE JSON line (<= 2 KB) on rank 0 - the LAST line of
stdout - with the contract's keys plus
  roofline          the dominant kernel timed INSIDE the replayed graph (device wall-clock stamps, hm_sil_timestamps), its
                    algorithmic bytes / that time / 8 TB/s (live); `traffic` / `valu_frac` from the committed PMC passes
                    under profiles/ (tools/pmc_loop.sh) when they were measured on the same shapes (`traffic_source` says so);
  cpu_baseline      the oracle loop timed on this host (bounded sample, rank 0 at N = 1 only);
  steady_state      one number: the same fit past iteration 400;
  multi_clip        one number: BASELINE cfg4 in miniature, `--multi-clip` clips per GPU as ONE clip batch;
  ranks / per_rank_its   what the process group reports and every rank's own rate (N > 1).
Everything else (per-kernel tables, notes, and with `--parity` the HIP-vs-oracle legs of bench_parity.py) goes to
gpurun_out/bench_detail.json and stderr.  The default run finishes in about a minute.
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
      k_bwd_lines : four 1-bit planes + dimg read; per-line records {64 mask bits, sources before them} (16 B per 64 samples,
                    four (plane, orientation) combinations) and per-line summaries (16 B per line and orientation) written;
                    its work-list blocks read faces + boxes + owned flags (46 B per face) and write, per face, either its
                    64-byte record (+ 4 B first item) or the 48 bytes of its zero gradient: 56 B on average.  (Round 4 left
                    the records, summaries and the work list's writes out - 23.8 MB for cfg2 - and read the PMC traffic of
                    the pose initialisation against that as "4.6 x": VERDICT r4.  The per-line SOURCE arrays - 12 B per
                    source and orientation - are data dependent and still not counted: ~0.3 MB in the steady state of a
                    fit, ~100 MB per launch for 500 candidate poses far from their mask.)
      k_bwd_sweep : face records of the work list (64 B + 4 B first item, bound: every face), index map, per-line records
                    read; per-face corner gradients (6 doubles: exact sums) written.  (Source arrays as above.)
    noaa (the pose initialisation's fused loop: rendering without anti-aliasing, per-sample masked L2 against ONE binary mask,
    hm_sil_fwd mask_shared = 3 + hm_sil_bwd mode 5): no per-sample image leaves the raster - the index map, the pooled
    silhouette and the bit-planes are its outputs - and the line expansion reads no gradient (it is -1 / +1 by plane)."""
    is_ = 2 * S
    is2 = is_ ** 2
    lines_out = 4 * is_ * (is_ // 64) * 16 + 2 * is_ * 16 + F * 56        # line records + summaries + work list, per frame
    if noaa:
        return {"k_raster_fwd": B * (F * (36 + 8 + 5) + is2 * 4 + S * S * 4 + 5 * is2 // 8),
                "k_bwd_sweep": B * (F * (64 + 4) + is2 * 4 + is2 + F * 48),
                "k_bwd_lines": B * (4 * is2 // 8 + is2 + F * (36 + 8 + 2) + lines_out)}
    return {"k_raster_fwd": B * (F * (36 + 8 + 5) + is2 * 4 + 4 * S * S * 4 + 5 * is2 // 8),
            "k_bwd_sweep": B * (F * (64 + 4) + is2 * 4 + is2 + F * 48),
            "k_bwd_lines": B * (4 * is2 // 8 + S * S * 4 + is2 + F * (36 + 8 + 2) + lines_out)}


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
    ppath = next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_poseinit.json", "r05_pmc_poseinit.json", "r04_pmc_poseinit.json"))
                  if os.path.exists(p)), None)
    if ppath:
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
                       f"{steps}-step fit's hipGraph (hm_sil_timestamps) - ONE loop over all {n} candidates (po._fused_loop), its "
                       f"launches alone on the GPU; the timed fits run through the resident fitter, which walks the candidates as "
                       f"`candidate_groups` loops side by side (same kernels over a third of the candidates each, overlapping)",
                traffic_source=f"profiles/{os.path.basename(ppath)} (committed rocprofv3 --pmc passes, tools/pmc_poseinit.sh)" if per[dom].get("traffic_bytes") else None)
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
                                 "candidate_groups": next((f.parts for f in po._FITTERS.values()), 1),
                                 "pose_steps_per_s_by_loop": {k: n * steps / v for k, v in loops.items()},
                                 "cold_fit": (dict(seconds_per_fit=cold, pose_steps_per_s=n * steps / cold) if cold else None)},
                      "best_iou": float(iou.max()), "seconds_per_fit": el, "roofline": roof, "cpu_baseline": cpu})


_REAL_STDOUT = None
MAX_LINE_BYTES = 2048          # the driver parses the LAST stdout line; round 4's 23 KB line did not parse


def _quiet_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes its version banner to fd 1 when
    a process group comes up), so everything but the final line is routed to stderr."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _num(x, digits=5):
    """floats of the line at `digits` significant digits (the full values go to the detail file)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "valu_frac", "traffic_source")
LEG_KEYS = ("cfg2_depth", "cfg3")      # one-number legs of the default run: BASELINE configs[1] as worded (with the depth term), configs[2]
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def compact_line(full):
    """The ONE stdout line (<= MAX_LINE_BYTES): the contract's keys, `roofline` and `cpu_baseline` as flat objects, and at most
    one-number summaries of the other legs.  `full` is the complete record (per-kernel tables, per-step traces, notes) - it
    goes to the detail file and stderr.  tests/test_cabi_and_host.py builds a line from canned numbers and checks the bound."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(full.get("config") or {})
    if len(str(cfg.get("workload", ""))) > 300:
        cfg["workload"] = str(cfg["workload"])[:297] + "..."
    line["config"] = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool)) or v is None}
    r = full.get("roofline")
    line["roofline"] = None if not r else {k: r.get(k) for k in ROOFLINE_KEYS if k in r}
    if line["roofline"] and r.get("whole_iteration"):
        line["roofline"]["whole_iteration_frac"] = r["whole_iteration"].get("frac")
    if line["roofline"] and r.get("raster_stage_8d"):
        line["roofline"]["stage_frac_8d"] = r["raster_stage_8d"].get("frac")
    c = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not c else {k: c.get(k) for k in CPU_KEYS if k in c}
    if line["cpu_baseline"] and len(str(line["cpu_baseline"].get("sample", ""))) > 160:
        line["cpu_baseline"]["sample"] = str(line["cpu_baseline"]["sample"])[:157] + "..."
    for k in ("clips_per_s", "clips_per_s_hbm_frac", "first_loss", "final_loss", "ranks", "backend", "per_rank_its",
              "shared_scale_final", "replicas_identical", "best_iou", "seconds_per_fit", "parity_ok", "parity_vs", "detail"):
        if full.get(k) is not None:
            line[k] = full[k]
    for k in LEG_KEYS:
        if full.get(k):
            line[k] = {kk: full[k][kk] for kk in ("value", "fit400", "ms_per_step", "dominant_kernel_us") if kk in full[k]}
    for k in ("steady_state", "multi_clip"):
        leg = full.get(k)
        if leg:
            line[k] = {kk: leg[kk] for kk in ("value", "unit", "ms_per_step", "ms_per_round", "clips_per_gpu", "frac") if kk in leg}
            if k == "steady_state" and leg.get("roofline"):
                line[k]["dominant_kernel_us"] = leg["roofline"].get("avg_launch_us")
                line[k]["frac"] = leg["roofline"].get("frac")
            if k == "multi_clip" and leg.get("roofline"):
                line[k]["frac"] = leg["roofline"].get("frac")
    line = _num(line)
    s = json.dumps(line)
    if len(s) > MAX_LINE_BYTES:           # never let an over-long line out again: drop the optional tail, keep the contract
        for k in ("per_rank_its", "clips_per_s_hbm_frac", "first_loss", "final_loss", "multi_clip", "steady_state") + LEG_KEYS:
            line.pop(k, None)
            if len(json.dumps(line)) <= MAX_LINE_BYTES:
                break
    return line


def emit(full):
    """Writes the complete record to the detail file (gpurun_out/bench_detail.json unless $HOMAN_BENCH_DETAIL names another
    path) and to stderr, and the compact line - the LAST line of stdout - to the real stdout."""
    path = os.environ.get("HOMAN_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full = dict(full, detail=os.path.relpath(path, ROOT))
    except OSError as e:                  # a read-only tree must not cost the line
        sys.stderr.write(f"bench.py: detail file not written ({e})\n")
    sys.stderr.write("bench detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    out = json.dumps(compact_line(full))
    assert len(out) <= MAX_LINE_BYTES and "\n" not in out, len(out)
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (out + "\n").encode())


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


_T0 = time.perf_counter()


def _leg(name):
    sys.stderr.write(f"[bench {time.perf_counter() - _T0:6.1f} s] {name}\n")
    sys.stderr.flush()


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
    ap.add_argument("--steady", type=int, default=1000,
                    help="steady_state leg after the headline: the same fit continued to iteration >= 400, then this many "
                         "timed iterations (BASELINE cfg2 is a 400-step fit; a short --steps/--warmup headline times the first, "
                         "heavier iterations); 0 = skip")
    ap.add_argument("--legs", default="cfg2_depth,cfg3",
                    help="one-number legs after the headline (default run only: one rank, cfg2 without --depth / --step2): "
                         "cfg2_depth = BASELINE configs[1] as worded, with the ordinal depth term; cfg3 = configs[2] (step-2 losses). "
                         "Each: a fresh 400-step fit, then 300 timed iterations of its steady state.  '' = none")
    ap.add_argument("--parity", action="store_true",
                    help="opt-in: the parity and end-to-end legs of bench_parity.py (cfg1 x --parity-seeds, --lockstep teacher-forced "
                         "steps, --freerun free-running steps, --e2e-clips through ClipFitter).  They are what pytest -m gpu "
                         "asserts on (tests/test_parity_gpu.py, test_lockstep_gpu.py); their records go to the detail file, "
                         "the line gets `parity_ok`")
    ap.add_argument("--parity-seeds", type=int, default=5)
    ap.add_argument("--lockstep", type=int, default=24)
    ap.add_argument("--freerun", type=int, default=100)
    ap.add_argument("--e2e-clips", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0,
                    help="seconds of CPU work of the cpu_baseline leg (rank 0, N = 1 only)")
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
    _leg("import torch")
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

    _leg("synthetic clip")
    mano = synthetic_mano(0)
    sil_fn, hand_fn = synth.hip_clip_fns(mano)
    if args.shared_scale:
        return bench_shared_scale(args, rank, world, backend, mano, sil_fn, hand_fn)
    clip = synth.make_clip(seed=rank, frames=args.frames, rend_size=args.size, image_size=args.size, obj="bottle",
                           silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    lw = dict(synth.STEP2_LOSS_WEIGHTS if args.step2 else synth.STEP1_LOSS_WEIGHTS)
    if args.depth:
        lw["lw_depth"] = 1.0
    _leg("model + stepper (calibration, graph capture)")
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

    per_rank = []

    def timed(n, keep=None):
        barrier()
        t0 = time.perf_counter()
        stepper.run(n)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0            # this rank's own time (before it waits for the others)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            dev = "cuda" if backend == "nccl" else "cpu"
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
            if keep is not None:
                o = torch.tensor([own], dtype=torch.float64, device=dev)
                g = [torch.zeros_like(o) for _ in range(world)]
                dist.all_gather(g, o)
                keep.extend(n / float(x.item()) for x in g)
        elif keep is not None:
            keep.append(n / own)
        return el

    _leg(f"headline: {args.steps} timed iterations")
    elapsed = timed(args.steps, per_rank)
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
        for cand in PMC_FILES:
            ppath = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(ppath):
                pj = json.load(open(ppath))
                want = dict(frames=B, rend_size=S, faces=int(F), step2=bool(args.step2))
                if args.depth:
                    want["depth"] = True
                if pj.get("shape") == want:
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
                    # VALU issue roof.  tools/valu_ceiling.hip measured what a wave64 instruction costs a SIMD (profiles/
                    # r06_valu_ceiling.json): ~2.3 cycles for fp32 / integer adds, multiplies and bit ops, ~4.2 for selects, compares,
                    # shifts, three-operand integer ops, conversions, DPP and doubles - neither the guide's flat 2 nor the counters' flat
                    # 4 (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.02 quad-cycles).  Priced with the kernel's STATIC opcode mix
                    # (tools/valu_mix.py, at 4 waves per SIMD) that is `cyc` cycles per instruction; valu_frac_4cyc is round 5's figure.
                    cyc = VALU_MIX_CYCLES.get(name, 3.4)
                    rec["valu_wave_instr"] = c["SQ_INSTS_VALU"]
                    rec["valu_cycles_per_instr"] = cyc
                    rec["valu_frac"] = c["SQ_INSTS_VALU"] * cyc / (sec * 1024 * 2.4e9)
                    rec["valu_frac_4cyc"] = c["SQ_INSTS_VALU"] / (sec * 1024 * 2.4e9 / 4)
                    if c.get("SQ_ACTIVE_INST_VALU"):
                        rec["quad_cycles_per_valu_instr"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
            per[name] = rec
        dom = max(per, key=lambda k: per[k]["avg_launch_us"])
        tot = algorithmic_bytes(B, S, F, V, args.step2)["total"]
        # the STRICT reading of SURVEY 8(d): its whole raster stage (134.7 MB at cfg2: every logical tensor once per consumer)
        # over the three kernels' summed durations - the per-kernel models above also count the kernels' own intermediates
        # (line records, work list) and sum to ~18 % more than this
        stage_us = sum(per[k]["avg_launch_us"] for k in per)
        stage = algorithmic_bytes(B, S, F, V, args.step2)["raster"]
        stage_rec = dict(algorithmic_bytes=stage, kernels_us=stage_us, achieved_GBps=stage / stage_us / 1e3,
                         frac=stage / stage_us / 1e3 / 8000.0)
        return dict(bound="hbm", raster_stage_8d=stage_rec, kernel=dom, achieved=per[dom]["achieved_GBps"], peak=8000.0, unit="GB/s",
                    frac=per[dom]["achieved_GBps"] / 8000.0, traffic=per[dom].get("traffic_bytes"),
                    avg_launch_us=per[dom]["avg_launch_us"], valu_frac=per[dom].get("valu_frac"),
                    timing=f"device wall clock stored by every workgroup at entry and exit (earliest start to latest end) in "
                           f"{reps} more replays of the timed hipGraph (hm_sil_timestamps); same launches, same overlap",
                    # `achieved` / `avg_launch_us` are LIVE (this run); `traffic` / `valu_frac` come from the committed PMC passes
                    traffic_source=(f"profiles/{psrc} (committed rocprofv3 --pmc passes, not this run)"
                                    if per[dom].get("traffic_bytes") else None),
                    kernels=per,
                    whole_iteration=dict(algorithmic_bytes=tot, achieved_GBps=tot * its_per_s / 1e9,
                                         frac=tot * its_per_s / 8.0e12))

    roof = steady = None
    if fused:
        _leg("roofline stamps")
        r = stamp_roofline(args.steps / elapsed, stamp_reps) if rank == 0 else stepper.run(stamp_reps)
        roof = r if rank == 0 else None
        if args.steady > 0:
            # the steady state of the same fit (BASELINE cfg2 is a 400-step fit: iterations 5-25, which short driver flags
            # time, are its heaviest - the band where render and target disagree is still wide, the sweeps see 5-10 M pairs
            # instead of 1.4 M): continue to iteration >= 400, then time `--steady` more
            _leg(f"steady state: {args.steady} timed iterations")
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
        _leg(f"multi_clip: {args.multi_clip} clips as one batch")
        C, msteps = args.multi_clip, min(max(args.steps, 100), 200)
        models = []
        for i in range(C):
            ci = synth.make_clip(seed=1000 + 100 * rank + i, frames=args.frames, rend_size=args.size, image_size=args.size,
                                 obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
            models.append(build_model(copy.deepcopy(ci["person_parameters"]), copy.deepcopy(ci["object_parameters"]),
                                      objvertices=ci["objvertices"], objfaces=ci["objfaces"], camintr=ci["camintr"],
                                      optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                                      sync_metrics=False, ordinal_depth=args.depth))
        # (the per-GPU worker of cfg4, dist.optimize_clip_shard, drives its shard through ShardStepper: clips of one shape as clip
        #  batches - two batches side by side for a step-1 shard of one shape, one batch otherwise)
        from homan_amd.jointopt import ShardStepper
        bst = ShardStepper(models, lw, 1e-2, msteps + 10)
        nbatches = len(bst.steppers)
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
                     clip_batches=nbatches,
                     note="the shard of a GPU through ShardStepper (dist.optimize_clip_shard's engine): clip batches - ONE launch per "
                          "kernel over the clips of a batch, one hipGraph per batch and iteration, the batches' graphs replayed side by "
                          "side; max over ranks")
        del bst, models

    legs = {}
    if fused and world == 1 and not args.step2 and not args.depth and args.legs:
        from homan_amd import lib as hlib
        import ctypes
        for name in [x for x in args.legs.split(",") if x in LEG_KEYS]:
            _leg(f"leg {name}")
            lwl = dict(synth.STEP2_LOSS_WEIGHTS if name == "cfg3" else synth.STEP1_LOSS_WEIGHTS)
            dep = name == "cfg2_depth"
            if dep:
                lwl["lw_depth"] = 1.0
            ml = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                             objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                             optimize_mano=True, image_size=args.size, mano_model=mano, rend_size=args.size,
                             sync_metrics=False, ordinal_depth=dep)
            lsteps, lreps = 300, 20
            sl = FusedStepper(ml, lwl, 1e-2, 400 + lsteps + lreps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sl.run(400)                       # BASELINE's wording of these configs: a 400-step fit - timed whole (`fit400`) ...
            torch.cuda.synchronize()
            fit400 = 400 / (time.perf_counter() - t0)
            t1 = time.perf_counter()          # ... then its steady state (`value`), like the steady_state leg of the headline
            sl.run(lsteps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            # the three heavy kernels of the silhouette chain inside the replayed graph (hm_sil_timestamps, as in stamp_roofline)
            L = hlib.lib()
            sctx = sl.model.sil_ctx
            ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
            us3, acc = (ctypes.c_float * 3)(), [0.0, 0.0, 0.0]
            for _ in range(lreps):
                hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "hm_sil_timestamps")
                sl.run(1)
                hlib.check(L.hm_sil_timestamps_read(ws, *dims, None, ctypes.cast(us3, ctypes.c_void_p), hlib.stream()),
                           "hm_sil_timestamps_read")
                acc = [a + float(u) for a, u in zip(acc, us3)]
            hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "hm_sil_timestamps")
            kus = dict(zip(("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep"), (a / lreps for a in acc)))
            dom = max(kus, key=kus.get)
            tot = algorithmic_bytes(B, S, F, V, name == "cfg3")["total"]
            legs[name] = dict(value=lsteps / el, fit400=fit400, unit="it/s", ms_per_step=1e3 * el / lsteps, dominant_kernel=dom,
                              dominant_kernel_us=kus[dom], kernels_us=kus, first_timed_iteration=400, steps=lsteps,
                              whole_iteration_frac=tot * (lsteps / el) / 8.0e12,
                              note="fresh 400-step fit of the same clip (fit400 = its rate, iterations 0-400), then its steady state "
                                   "(value, iterations 400-700); the silhouette chain's kernels "
                                   "stamped inside the replayed graph (the depth renders and SDF kernels are not stamped; with the depth term "
                                   "k_raster_fwd is the span of the SILHOUETTE render's workgroups inside the hm_sil_fwd_multi launch it "
                                   "shares with the object's depth render)")
            del sl, ml

    cpu = parity = e2e = None
    parity_ok = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _leg(f"cpu_baseline: oracle loop, {args.cpu_budget:.0f} s budget")
        cpu, evo_cpu = cpu_baseline(clip, lw, mano, args.cpu_budget, rend_size=S, image_size=S, ordinal_depth=args.depth)
        if args.parity:
            import bench_parity as bp
            _leg("parity legs (opt-in)")
            parity = dict(cfg2_first_steps=bp.trajectory_parity(evo, evo_cpu),
                          lockstep=(bp.lockstep_parity(mano, step2=args.step2, steps=args.lockstep, frames=B, size=S, clip=clip,
                                                       lw=lw, free_run=not args.depth, ordinal_depth=args.depth)
                                    if args.lockstep > 0 else None),
                          cfg1=bp.cfg1_parity(mano, seeds=list(range(args.parity_seeds))) if args.parity_seeds > 0 else None,
                          free_run=(bp.free_run_parity(mano, step2=args.step2,
                                                       steps=min(args.freerun, 40) if (args.step2 or args.depth) else args.freerun,
                                                       frames=B, size=S, clip=clip, lw=lw, ordinal_depth=args.depth)
                                    if args.freerun > 0 else None),
                          bar="north_star: 1e-4 relative on losses, 1e-3 mm on final vertices.  cfg1 / free_run compare FREE-running "
                              "trajectories, HIP fused loop vs the oracle's REPRODUCIBLE (written-out) loop: every parameter "
                              "bit-equal after every step.  cfg2_first_steps is the headline run against the cpu_baseline leg's "
                              "FAITHFUL oracle loop (torch Adam, autograd), which separates once a sample flips (DESIGN.md 2)")
            checks = []
            if parity["cfg1"]:
                checks.append(bool(parity["cfg1"]["all_within_bars"]))
            if parity["free_run"]:
                fr = parity["free_run"]
                checks.append(fr["first_step_over_tol"] is None and max(fr["final_vertex_diff_mm"].values()) < 1e-3)
            if parity["lockstep"]:
                checks.append(parity["lockstep"]["first_step_over_tol"] is None)
            parity_ok = all(checks) if checks else None
    if args.parity and rank == 0 and world == 1 and fused and args.e2e_clips > 0 and not args.depth:
        import bench_parity as bp
        _leg("end to end (opt-in)")
        del stepper
        e2e = bp.end_to_end_clips(mano, lw, clips=args.e2e_clips,
                                  clips_per_batch=max(1, min(args.multi_clip or 1, args.e2e_clips // 2)), steps=400, frames=B, size=S)
    if rank == 0:
        value = world * args.steps / elapsed
        line = {
            "metric": "optimisation iters/sec (30-frame 256^2 clip)", "value": value, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("cfg3" if args.step2 else "cfg2") +
                       f": 1 clip/GPU x {B} frames {S}x{S}, MANO hand + bottle ({F} faces), "
                       + ("step-2" if args.step2 else "step-1") + " losses" + (" + ordinal depth" if args.depth else "")
                       + ", fwd+bwd+Adam+logging per step from a hipGraph",
                       "frames": B, "rend_size": S, "faces": int(F), "clips_per_gpu": 1,
                       "loop": "fused" if fused else "autograd graph", "parallelism": f"{world} independent clips"},
            # BASELINE's second reading of the metric: clips/s (a clip = one 400-step fit) absolute and as the fraction of the
            # N x 8 TB/s HBM roof the SURVEY 8(d) byte model of those iterations amounts to
            "clips_per_s": value / 400.0,
            "clips_per_s_hbm_frac": algorithmic_bytes(B, S, F, V, args.step2)["total"] * value / (8.0e12 * world),
            "final_loss": evo["loss"][-1], "first_loss": evo["loss"][0],
            # what the process group itself says (N > 1: RCCL's rank count), and every rank's own rate over the timed region
            "ranks": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
            "per_rank_its": per_rank,
            "roofline": roof, "steady_state": steady, "cpu_baseline": cpu, "multi_clip": multi,
            "cfg2_depth": legs.get("cfg2_depth"), "cfg3": legs.get("cfg3"),
            "parity_ok": parity_ok,
            "parity_vs": ("oracle's reproducible (written-out) loop, end state" if parity_ok is not None else None),
            "final_loss_parity": parity, "end_to_end": e2e,
            "clips_per_s_end_to_end": e2e["clips_per_s_end_to_end"] if e2e else None,
        }
        _leg("done")
        emit(line)
    if world > 1:
        dist.destroy_process_group()


def _valu_mix_cycles():
    """mix-weighted cycles per wave64 VALU instruction of the three heavy kernels (profiles/r06_valu_mix.json, 4 waves per SIMD)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_mix.json")))
        return {k.split("<")[0]: v["mix_cycles_per_instr_w4"] for k, v in d.items() if "mix_cycles_per_instr_w4" in v}
    except (OSError, ValueError, KeyError):
        return {}


VALU_MIX_CYCLES = _valu_mix_cycles()
PMC_FILES = ("r06_pmc_loop.json", "r06_pmc_loop_cfg3.json", "r06_pmc_loop_depth.json", "r05_pmc_loop.json", "r05_pmc_loop_cfg3.json",
             "r04_pmc_loop.json", "r04_pmc_loop_cfg3.json")


if __name__ == "__main__":
    main()
