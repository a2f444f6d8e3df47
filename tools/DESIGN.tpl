# DESIGN — homan_amd: HOMan's joint-optimisation hot path on MI355X (gfx950)

Scope is exactly SURVEY.md §8 (the grading contract): rows (a)–(e), then the four "next" rows of §8f, each to the same
parity bar.  Everything else in the reference (detection, tracking, datasets, mask / hand-pose networks, dataset
evaluation) stays out of scope.  All `path:line` citations are relative to `/root/reference/`.  This file is the CURRENT
design and the current numbers; how the kernels got here - every measured step and every reverted experiment - is in
`EXPERIMENTS.md`.

## 0. §8 rows → where they live

| §8 row | What | Where |
|---|---|---|
| (a) a1 loop | `optimize_hand_object`, three Adam groups, weighting, logging | `homan_amd/jointopt.py` (the reference's module name: `optimize_hand_object`, `GraphStepper`, re-exports) over `homan_amd/loopcommon.py` (collation, Adam groups, fused Adam, device log), `homan_amd/fused.py` (`FusedStepper`: the iteration issued by per-chain builders - silhouette chain, hand forward, pair terms, depth terms, hand backward, object backward, join) and `homan_amd/shard.py` (`ShardStepper`, `ClipFitter`) (`mode="auto"` (default) = `"fused"` whenever `FusedStepper` accepts the configuration - its own guards decide - else `"graph"`; `"eager"` = reference loop verbatim; `GraphStepper` = the same autograd iteration in a hipGraph; `FusedStepper` = the iteration as a fixed C-ABI launch sequence on two (one clip) or three (clip batch) streams in a hipGraph, the benchmark path - one clip, a BATCH of equal-shaped clips (`homan_amd/clipbatch.py`), or through `ShardStepper` a shard of clips of any shapes, the shape groups replayed concurrently; `ClipFitter` = the dataset walk of `fit_vid_dataset.py:190-379` on RESIDENT steppers: a clip of a known shape is copied into the static buffers and the resident hipGraph replayed), `csrc/adam.hip` |
| (a) a2,a3,a7,a20 | `HOMan` module: parameter/buffer surface, `get_verts_*`, `forward` | `homan_amd/homan.py` |
| (a) a4–a6 | rot6d→R, rigid transform (+ mesh-detached twin) | `csrc/geometry.hip` (`hm_rigid_fwd/bwd`) |
| (a) a8 + N3 | `ManoModel.forward_pca` + MANO LBS | `csrc/mano.hip` (`hm_mano_fwd/bwd`), `homan_amd/manomodel.py`, `mano_assets.py` |
| (a) a9,a10,a13,a15 | pca / smooth / v2d / scale priors | `csrc/losses.hip` |
| (a) a11,a12 + N1 | silhouette renderer + masked MSE + IoU | `csrc/raster_{setup,fwd,lines,sweep,api}.hip` over `raster_common.h` / `raster_ws.h` (`hm_sil_fwd/bwd`) |
| (a) a14 | coarse interaction loss + gating + min-distance metric | `csrc/losses.hip` (`hm_inter_*`), `csrc/contact.hip` (`hm_nn_fwd`) |
| (a) a16,a17 + N2 | SDF collision loss | `csrc/sdf.hip` (`hm_collision_fwd`) |
| (a) a18 | contact loss (as executed, Appendix B.1) | `csrc/contact.hip` |
| (a) a19 | ordinal depth | `csrc/raster_depth.hip` (+ the depth output of `raster_fwd.hip`): depth image out of `hm_sil_fwd` (`pooled_depth`), `hm_depth_bwd`, `hm_ordinal_depth_fwd/bwd`. The reference call site is broken (`homan.py:506-507` raises `TypeError`, `lossutils.py:140` builds the accumulator with `torch.Tensor(0.0)`): the default still raises that `TypeError`; `HOMan(ordinal_depth=True)` opts into the loss the method describes, in all three loops - eager, graph and the fused launch sequence (`FusedStepper`, `lw_depth > 0`, one clip, one or - round 5 - two hands per frame: three layers, three pair terms, the scene's normaliser and the pairs' shares formed on the device; any render size since the rasteriser pads to a multiple of 32, the fused depth renders at `image_size % 32 == 0`). **Oracle-pinned only** (no reference output exists to pin against) |
| (b) boundary | Python surface + C ABI | `homan_amd/{homan,losses,lossutils,manomodel,jointopt}.py`, `include/homan_amd.h`, `INTEGRATION.md` |
| (c) oracle | CPU restatement + reference-generated goldens; the object's gradient chain and Adam also written out with order-independent sums (`oracle/objchain.py`, `oracle/csrc/objchain.c`, `oracle/adam.py`) | `oracle/`, `tools/refharness/`, `tests/golden/` |
| (d) measurement | bench, roofline, CPU baseline, rocprof | `bench.py`, `profiles/`, `tools/prof_summary.py` |
| (e) multi-GPU | clip sharding, clip batches per rank, heterogeneous shards, shared-scale all-reduce inside the fused loop | `homan_amd/dist.py`, `homan_amd/clipbatch.py`, `jointopt.ShardStepper`, `FusedStepper(shared_scale=True)`, `bench.py --gpus N [--shared-scale]`, `tests/test_dist_gloo.py` (CPU, oracle model), `tests/test_dist_gpu.py` (N ranks on one GPU through the fused loop, cfg5 at full size), `tests/test_clip_batch_gpu.py` |
| (f) rank 1 | object-pose initialisation (`pose_optimization.py:37-160,219-383`, `lib3d/optitrans.py:83-127`) | `homan_amd/pose_optimization.py` (`PoseOptimizer`, `find_optimal_pose`, and the clip-level `find_optimal_poses` of `:386-488` that `fit_vid_dataset.py:285-296` calls) over the same rasteriser run without anti-aliasing (`hm_sil_fwd` `alpha_full` with the masked L2 + IoU fused per sample, `hm_sil_bwd` modes 3 / 4 / 5); the default loop of `find_optimal_pose` is a fixed launch sequence without the autograd tape, run by a resident `PoseFitter` (`_FusedPoseLoop`: `hm_rigid_fwd` → `hm_offscreen_fwd` → raster → reduce → lines / sweeps → `hm_rigid_bwd_sil` → `hm_adam_step` → `hm_pose_keep_best`, one hipGraph), `mode="eager"` is the reference's loop verbatim; oracle `oracle/poseopt.py`; golden from the reference's own module (`tools/refharness/gen_goldens_poseinit.py`); `bench.py --pose-init` |
| (f) rank 2 | hand silhouette term (`losses.py:166-181`) + ordinal depth (`homan.py:384-419`, `lossutils.py:133-169`), both present-but-disabled upstream | `Losses.compute_sil_loss_hand` (per-hand ROI render, own keep-mask normalisation; the reference body cannot run past its first frame, built for the evident intent) ; ordinal depth = row a19.  Oracle-pinned |
| (f) rank 3 | textured / shaded renders for visualisation (`homan.py:168-219,510-628`, `visualize.py:44-128`, `meshutils.py:7-51`, `jointopt.py:158-176`) | rgb output of the rasteriser (`hm_shade_rgb`: NMR flat lighting on the forward's index map), `homan_amd/nmr.py` (`model.renderer`), `HOMan.render / render_gt / render_with_gt / render_limem / save_obj`, `homan_amd/{meshutils,visualize,trans3d}.py`, frames out of `optimize_hand_object`; oracle `oracle/nmr.py` (`lighting`, `shade_index_map`); `tests/test_render_gpu.py` |
| (f) rank 4 | checkpoint / evaluation hand-off (`fit_vid_dataset.py:365-372,322-338`, `postprocess.py:16-77`, `eval/pointmetrics.py:102-124`) | `homan_amd/checkpoint.py` (`joint_fit.pt` contract, files interchange with the reference's), `homan_amd/pointmetrics.py::get_inter_metrics` on `hm_collision_fwd` + `hm_collision_dist_values` (per-vertex penetration depths = `sdf_meta["dist_values"]`) |

## 1. The path and its boundary

Hot path = `homan.jointopt.optimize_hand_object` (`homan/jointopt.py:22-201`) → `HOMan.forward`
(`homan/homan.py:421-508`) → `homan/losses.py`, `homan/lossutils.py`, `homan/interactions/{scenesdf,contactloss}.py`,
`homan/utils/{geometry,camera,bbox}.py`, `homan/manomodel.py` → third-party leaves `neural_renderer`, `sdf`, `mano`.

Two boundaries are kept:

* **Python (what the reference's callers see).**  `homan_amd.HOMan` has the reference constructor keywords
  (`homan.py:27-60`), the same Parameter / buffer *names* (so the name-substring Adam grouping of
  `jointopt.py:128-151` and `load_state_dict(strict=False)` of a `joint_fit.pt` behave identically), the same
  `forward(loss_weights) -> (loss_dict, metric_dict)` keys / shapes (`()` except `loss_contact, loss_sil_obj, loss_inter`
  = `(1,)`), `get_verts_object()`, `get_verts_hand(detach_scale=False)`.  `homan_amd.optimize_hand_object` has the
  reference signature and return triple.  Extensions are keyword-only (`mano_model`, `rend_size`, `mode`).
* **C ABI (`include/homan_amd.h`, `libhoman_amd.so`).**  Plain pointers + sizes + `hipStream_t`, caller-owned buffers,
  error codes, no torch types.  The Python layer binds it with `ctypes` (`homan_amd/lib.py`); torch supplies device
  memory, streams, the autograd tape and `torch.distributed` only.  The silhouette backward leaves its per-(face, corner)
  gradients as doubles (`hm_sil_parts`) and takes, like `hm_rigid_bwd_sil*`, the grid of the order-independent sums
  (`sum_log2q`, section 2).  There is **no CPU fallback**: `lib.lib()` raises if
  the library is missing, `HOMan.__init__` raises without a GPU, and `homan_amd` never imports `oracle`.  Every entry point
  of the loop also exists as `hm_*_clips(..., clip_len[, out_stride])` (frame axis = several clips laid end to end, per-clip
  scalars and outputs as arrays) and the MANO pair as `hm_mano_{fwd,bwd}_rows` (a strided slice of the rows: hand `i` of
  two interleaved hands through the model of its side, `homan.py:343-358`, without a copy).

**What the default `mode="auto"` runs in the fused launch sequence** (everything a golden exists for): one hand, right or
left, or two hands per frame (one clip); `optimize_mano` on or off (`jointopt.py:36`: off is the reference function's own
default); `inter_type` `"centroid"` or `"min"` (one hand, one clip); a free or tied object scale; silhouettes at ANY
`rend_size` and any image size (sizes off the 32-pixel tile grid render on the next multiple with the first two rows of K
rescaled, masks padded with keep = 0 and the rasteriser's eps scaled - same rays, cropped outputs; the one difference of a
padded render: a face that leaves the image through its right / bottom border still lies inside the raster, so its edges
sweep where a native render of that size culls them); object meshes of any size (the metric-only search covers 4096
vertices, larger meshes take the full search; the contact scatter walks the object in ranges of 4096).  What falls back to
the graph loop (`HOMan.forward` + autograd captured in a hipGraph): `hand_proj_mode="ortho"` (section 7) and a free hand scale
(`optimize_mano_beta=False`).  Configurations the fused loop takes ONE clip at a time (two hands
per frame, `inter_type="min"`) run as one stepper per clip inside a shard, replayed side by side (`ShardStepper`, section 6).

Reference quirks reproduced on purpose (SURVEY Appendix B): `loss_contact ≡ mean 0.02·tanh(d_NN/0.02)`; `loss_inter` is
the un-normalised sum; `mano_rot`/`mano_trans` get gradients but are in no Adam group; `mano_betas` start at zero;
`int_scale_init` must be an `int`; `loss_sil_obj` divides by `Σkeep` of the whole clip, then by `B`; `lw_depth>0` raises
`TypeError`; `optimize_mano=False` has no `mano_trans`.  **Documented divergence:** `rot6d_to_matrix` takes the cross
product per row; the reference's dim-less `torch.cross` (`utils/geometry.py:26`) silently crosses along the batch axis
when the flattened batch is exactly 3 (goldens therefore never use 3 frames).

## 2. Oracle and parity status

`oracle/` is test infrastructure (only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` /
`final_loss_parity` legs import it).

* **Composition layer — PINNED.**  `tools/refharness/` imports the reference's own `homan/{homan,losses,lossutils,
  jointopt}.py`, `interactions/*`, `utils/*` *in place* (no copy, no bytecode) with `.cuda()` neutralised and the oracle
  leaves injected as `neural_renderer`, `sdf`, `mano.model`, `libyana.*`; `gen_goldens.py` runs the reference's
  `HOMan.forward` / `optimize_hand_object` on seeded synthetic clips and writes inputs + `loss_dict`, `metric_dict`, every
  Parameter gradient, the parameters after 5 reference steps with the reference's forward / backward THERE, 6-20-step
  `loss_evolution`, final parameters to `tests/golden/*.npz` (8 cases: step-1, step-2, free scale, `optimize_mano=False`,
  two hands step-1 / step-2, a lone left hand, `inter_type="min"`) + the pose-initialisation golden.
  `tests/test_oracle_golden.py` pins `oracle/model.py` + `oracle/jointopt.py` to them (losses 2e-5, gradients 5e-5 of max,
  vertices 3e-7 m, trajectories 5e-4) and re-runs the generator against `/root/reference` to check the committed file.
  Round 5 added the BASELINE-sized cases: `ref_cfg1_cube_b10_s128` (configs[0], the configuration the reference's CPU path is
  defined on: forward / backward, the 5-step pin and the reference loop's whole 100-step `loss_evolution`) and
  `ref_cfg2_bottle_b30_s256` / `ref_cfg3_bottle_b30_s256` (configs[1] / [2]: one forward / backward of the reference's `HOMan`
  at 30 frames x 256^2).  Oracle (both forms) and HIP model are pinned against them like against the small ones; the fused
  loop's cfg1 fit follows the reference's `loss_evolution` (`tests/test_model_gpu.py::test_cfg1_full_fit_follows_the_reference_loop`:
  tight for the 5-6 steps the two runs see the same coverage, then 35 % + 5 % of a term's first value, final total within 10 %;
  measured: total within 5.6 % at every one of the 100 steps, final 0.00672 vs 0.00656).
* **What "bit-equal" is measured against.**  Two forms of the oracle exist.  The FAITHFUL form (`oracle.model.REFERENCE_FORM`,
  autograd + `torch.optim.Adam`: the reference's literal expressions) is what the goldens pin and what the teacher-forced
  lock-step tests compare with, at tolerances (losses 1e-5, gradients 2e-5 / 5e-4).  The WRITTEN-OUT form (the same
  mathematics with every reduction and transcendental in one stated order, `oracle/{objchain,handchain,depthchain,posechain}.py`,
  `oracle/csrc/{objchain,lbs_exact}.c`) was written next to the kernels, and the "bit-equal after every step" statements of
  this file are HIP against THAT form.  The bridge between the two forms is tolerance-based (`tests/test_objchain.py`: 2e-5 of
  the largest gradient entry, 2e-6 on Adam; both forms against the goldens).  Independence from the reference therefore rests
  on the goldens and the faithful-form lock-step legs (50 steps each for cfg2 and cfg3, at full size), not on the bit-equality.
* **Leaves — PARITY UNPINNED.**  `neural_renderer`, `sdf` (hassony2/multiperson @ HEAD), `mano` (hassony2/MANO @ HEAD),
  `libyana` (@ HEAD) are un-vendored, un-pinned, absent from `/root/reference`, and the reference holds no tests or golden
  vectors for them; `MANO_RIGHT.pkl` is licensed and absent.  `oracle/csrc/nmr_raster.c`, `oracle/csrc/sdf.c`,
  `oracle/lbs.py`, `oracle/yana.py` restate the *published* algorithms (Kato et al. 2018 NMR; SDFGen-style voxel SDF; smplx
  LBS) and write down every convention they fix.  The reference build is **unbuildable here** (Python only; its native leaves
  are CUDA extensions whose sources are not in the tree), so there is no `oracle/_ref`.  `tests/test_known_answers.py`
  anchors oracle AND kernels to analytic truths any correct implementation satisfies (full-screen quad, half-plane edge in
  quarter steps, fill_back, cube SDF `h − max|x_i|`, LBS at rest, coincident contact = 0, pseudo-gradient magnitude of a
  single edge in float64).
* **Evaluation-order choices of the oracle** (same mathematics; upstream's own rounding - nvcc FMA contraction, cuBLAS - is
  not reproducible on any CPU): depth `z = Σw / Σ(w_k·(1/z_k))` with one division per sample; and, since round 3, the rotation
  (`rot6d_to_matrix`), the rigid transform `(s·v)·R + t` and the projection written out OPERATION BY OPERATION with a
  correctly rounded square root.  Measured reason: `torch.sqrt` on contiguous fp32 goes through MKL's vector math and is not
  correctly rounded (0.65 % of inputs one ulp off, depending on layout), the CPU `torch.matmul` is an FMA chain for V ≥ 45
  and a plain loop below, `F.normalize` / `torch.cross` round differently with the batch size - the reference's literal
  expressions give different last bits on different hosts, and one ulp in R moves a vertex across a sample centre.  The
  kernels follow the written-out order with `-ffp-contract=off` on both sides.

HIP ↔ oracle parity is exact where the domain is discrete and tolerance-bound where it is fp32 summation order:

| Quantity | Bar | Test |
|---|---|---|
| rotation, rigid transform, projected NDC faces | **bit-exact** | `test_ops_gpu.py::test_rigid_transform_and_grads`, `test_raster_gpu.py::test_projection_matches_oracle` |
| face-index map (B,2S,2S), pooled silhouettes | **bit-exact** (from the PARAMETERS, end to end) | `test_face_index_map_bit_exact`, `tests/test_lockstep_gpu.py` (0 flipped samples along 50 cfg2 / 30 cfg3 steps) |
| SDF inside/outside masks AND distance grids; MANO vertices (right, left, two hands) | **bit-exact** | `test_ops_gpu.py::test_collision_vs_oracle`, `::test_mano_lbs_and_grads`, `test_model_gpu.py` |
| every `loss_dict` entry vs reference goldens | **1e-4 relative** (BASELINE north_star) | `test_model_gpu.py::test_forward_matches_reference_goldens`, `test_pinned_step_matches_reference` |
| parameter gradients vs goldens | **5e-5** of max per tensor (2e-3 before round 3) | same |
| NMR pseudo-gradient per vertex | 2e-5 of max | `test_pseudo_gradient_matches_oracle` |
| **every step of a full-size trajectory, teacher-forced, vs the FAITHFUL (autograd) oracle** | losses 1e-5 (measured 3.3e-7, `loss_collision` 1.4e-7), gradients 2e-5 (cfg2) / 5e-4 (cfg3: the reference's nearest-vertex search in near-ties) of max, object AND hand vertices bit-equal, 0 flipped samples (silhouette and both depth renders) | `tests/test_lockstep_gpu.py` |
| C clips in one launch per kernel vs C solo runs; shape groups of a shard vs solo runs; 2 ranks vs one 2-clip batch | **bit-exact** (rows, final parameters) | `tests/test_clip_batch_gpu.py`, `tests/test_dist_gpu.py` |
| fused launch sequence vs `HOMan.forward` + autograd | losses 2e-6, gradients 2e-5 of max, every golden incl. two hands / left / `min` / `optimize_mano=False` | `test_fused_step_equals_autograd_path` |
| pose initialisation vs the reference module's golden (reference form: `torch.matmul`) | mask loss within the number of samples that differ (<= 3 per pose), gradients 2e-3 of max | `tests/test_poseinit.py` |
| pose initialisation vs the oracle with the written-out transform | coverage **bit-exact** (0 flipped samples), mask loss equal, gradients 5e-5 of max (autograd side) | `test_hip_poseinit_coverage_bit_exact_and_gradients_vs_written_out_oracle` |
| pose initialisation, a WHOLE free-running fit vs the oracle's written-out loop (`oracle/posechain.py`) | every candidate's rotation / translation, the per-candidate losses, the best-ever pose **bit-equal** | `test_fused_poseinit_fit_bit_equal_with_the_written_out_oracle` |
| every stage of the gradient chains at identical parameters vs the oracle's WRITTEN-OUT chains (unit gradients, interaction records, nearest-vertex picks, contact / collision / depth gradients, model-space gradient) and all parameter gradients | **bit-equal** | `tests/test_handchain_gpu.py` |
| FREE-running fit, HIP loop vs the oracle's reproducible loop: cfg1; cfg2; cfg2 + depth; cfg3; free / tied object scale; two hands; `optimize_mano=False` | EVERY parameter **bit-equal after every step** (400 steps at full size for cfg2 / cfg2 + depth / cfg3), losses 1e-4 (measured 3.6e-7), final vertices 0.0 mm | `tests/test_parity_gpu.py`, `tests/test_handchain_gpu.py`, `bench_parity.free_run_parity`, `profiles/r06_freerun_{cfg2,cfg2_depth,cfg3}_400.json` |
| a stream of clips through resident steppers vs fresh fits | **bit-exact** (parameters, vertices, loss_evolution) | `tests/test_clip_fitter_gpu.py` |

**Final-loss / final-vertex parity (the second half of BASELINE's metric): met, free-running.**  The hard rasteriser makes the
silhouette loss piecewise constant in the pose, and Adam's normalised step turns a last-bit difference in a gradient into a
flipped sample a few steps later: rounds 1-3 measured the HIP loop and the CPU loop centimetres apart after 100 steps although
every single step agreed to 2e-6 (the CPU oracle against ITSELF from inputs 1e-7 m apart still ends 4-14 mm apart:
`final_loss_parity.cfg1.cpu_vs_cpu_control`).  The only cure is to leave no last bit open, and on the step-1 loss sets the
object's chain is closed on itself (`homan/homan.py:482-490`: `loss_inter` sees the object detached), so it can be done:

* every REDUCTION on that chain - the sweeps' per-(face, corner) sums, the vertex gather, the 13 per-frame sums of the rigid
  backward - rounds its addends to multiples of 2^-44 (`(x + M) - M` in double, `hm_quant`) and adds them in double: exact while
  |sum| < 512, hence a function of the SET of addends, whatever lanes, units, atomics or workgroups formed it (include/homan_amd.h,
  "ORDER-INDEPENDENT SUMS"; the pose initialisation's unnormalised loss uses 2^-24);
* every per-term OPERATION is IEEE in one written order: `c = num / (p1 - d0)`, `k = (c * 2) / is`, `dist = k * (d1 - cross) +- eps`,
  `term = diff / dist` (no `v_rcp_f32`); the projection backward, the rot6d backward and Adam element by element; Adam's bias
  corrections by square-and-multiply in double (a libm `pow` differs in the last bit from host to host);
* the oracle evaluates the same chain the same way (`oracle/objchain.py` + `oracle/csrc/objchain.c` + `oracle/adam.py` =
  `oracle.jointopt.reproducible_step`: forward + autograd as always, then the object's pose gradients REPLACED by the written-out
  chain, then the written-out Adam).  Same mathematics as autograd + `torch.optim.Adam`: `tests/test_objchain.py` holds the
  two within fp32 rounding (2e-5 of the largest gradient entry, 2e-6 on Adam), the faithful forms stay pinned to the
  reference's goldens, and the oracle's result no longer depends on its thread count (1 vs 4 threads: bit-identical);
* the HAND's chain needs no new sums - `csrc/mano.hip` and `csrc/pair_bodies.h` were deterministic already (fixed vertex chunks,
  DPP trees, chunk records added in chunk order, no data atomics) - but one DEFINED order on both sides: `hm_sincos` (argument
  reduction + the fdlibm kernel polynomials in double, rounded to fp32; libm's and OCML's `sinf` differ in the last bit) is shared
  by the kernels and the oracle, the MANO layer's forward is written out in the kernels' order (`oracle/csrc/lbs_exact.c`:
  `orc_mano_forward`; joint regressor folded in double, so no host BLAS is involved) and so is the backward of the whole hand
  side (`orc_hand_chain`: per-vertex terms, the rigid backward's 12 sums, skinning / blend-shape / chain / Rodrigues / PCA
  backward with the kernels' reduction trees; `orc_v2d_unit_grad`, `orc_inter_rec`; `oracle/handchain.py`).  The torch
  restatement of the layer (`oracle/lbs.py`, `oracle.model.REFERENCE_FORM`) stays: both forms are pinned to the reference's
  goldens (`tests/test_oracle_golden.py`, both parametrisations) and to each other within an ulp;
* the step-2 PAIR TERMS the same way (`oracle/handchain.py` `pair_terms`, C in `oracle/csrc/lbs_exact.c`): the nearest object
  vertex of every hand vertex by differenced coordinates, ties to the lowest index (`orc_nn_search` <-> `nn_full_body`); the
  contact term with a shared `hm_tanh` (ln2 reduction + the fdlibm exp kernel in double; OCML's and glibc's `tanhf` differ in the
  last bit), element-wise on the hand, on the object the kernels' 2^-44 fixed-point sum of the picks' gradients - an integer sum,
  exact in any order (`orc_contact_grads` <-> `k_contact_both`); the collision term's gradient as the eight-corner trilinear
  expression of `k_sdf_sample` on the oracle's own SDF grid (`orc_sdf_sample_grad`; the grids - a min over triangles of one
  shared point-triangle routine, signs by ray parity - were bit-equal already).  One deviation from the reference's arithmetic
  is inherited from the kernels and stated: the reference ranks neighbours by `|a|^2 + |b|^2 - 2ab` (contactloss.py:60-79), whose
  rounding (~4e-8 m^2 at |a|^2 ~ 0.36 m^2) names another neighbour when two object vertices are within that of each other; the
  faithful form stays in `oracle/model.py`, and `tests/test_objchain.py` bounds the difference (picks at the same distance
  to 1e-7 m^2, gradients within 3e-4);
* the ORDINAL DEPTH term (`oracle/depthchain.py`): both meshes rendered at the full-image camera by the oracle's rasteriser
  (owner maps and z-buffers bit-equal with the kernels'), the four samples of a pixel pooled as `((s00 + s01) + s10) + s11`, the
  per-pixel gradient with counts as normalisers and a shared `hm_sigmoid`, NMR's depth-map backward per (face, winding) with the
  kernel's walk of the face's sample box (64 lanes in strides, then the wave tree - `orc_depth_bwd_faces`), the vertex gather in
  adjacency order, the results as one more term of the two rigid / MANO backward passes;
* a FREE OBJECT SCALE (cfg5's option): the interaction term then reaches the object's vertices, the scale's gradient is the
  frames' exact partial sums through one block sum plus the prior's term (`oracle/objchain.py`); tied across the clips of a
  rank, the clips' gradients meet in one more block sum and every replica takes the sum
  (`oracle.jointopt.reproducible_step_shared_scale`).

Measured (`final_loss_parity.{cfg1, free_run}` of the bench line, `tests/test_parity_gpu.py`, `tests/test_handchain_gpu.py`,
`profiles/r06_freerun_{cfg2,cfg2_depth,cfg3}_400.json`, regenerated with this round's kernels): EVERY parameter - `rotations_object`, `translations_object`, `rotations_hand`,
`translations_hand`, `mano_pca_pose`, `mano_rot`, `mano_betas`, `mano_trans` - is BIT-EQUAL between the two free-running loops
after every step: cfg1 100 steps x 5 seeds; cfg2, cfg2 + ordinal depth term and cfg3 (step-2: collision + contact) at full size
over 400 steps (`r06_freerun_cfg2_400.json`, `r06_freerun_cfg2_depth_400.json`, `r06_freerun_cfg3_400.json`:
`all_params_bit_equal_all_steps: true`); the step-2 set with a free object scale over 12 steps (`tests/test_handchain_gpu.py`); final
vertices 0.0 mm apart for the object AND the hand, every logged loss within 3.4e-7 at every step (bar 1e-4; the logged VALUES are
parallel float sums and keep their rounding, the trajectory does not see them).  Until the hand's chain was written out (first
half of this round) the hand separated around step 170-180 of the cfg2 clip - 0.14 mm at step 400 - exactly where the CPU loop
separates from ITSELF when its hand translations start 1e-7 m apart (`r04_control_cfg2_400.json`: 0.85 mm): Adam at 10 x lr on
the MANO parameters amplifies any difference, so only a chain without any could close it.  Two hands per frame (right + left,
rows interleaved, the step-2 set with its three collision scenes, fixed or free scale) are written out as well and bit-equal over
10 free-running steps (`tests/test_handchain_gpu.py::test_two_hands_bit_equal`).  So are `optimize_mano=False` (the hand mesh an input, its rigid
pose optimised), `inter_type="min"` (the closest vertex pair's pull on the hand's rigid pose) and - round 5 - two hands WITH the
ordinal depth term (three layers, three pairs, one normaliser: `oracle/depthchain.py::depth_vertex_grads_layers`; the pooled depth
images, pair counts, per-layer gradient images, vertex gradients, parameter gradients and 10 free-running steps with the step-1 and
the step-2 weights, `::test_two_hands_with_depth_term_bit_equal`).  Round 6 wrote out the last refused combination, `inter_type="min"` with a free
object scale (the closest pair's pull then reaches the object's vertex j* too: `oracle/objchain.py`; against autograd on the CPU,
`tests/test_objchain.py`, and bit-equal with the fused loop over 8 free-running steps on both loss sets,
`tests/test_handchain_gpu.py::test_inter_type_min_with_a_free_object_scale_bit_equal`).  What the written-out hand chain still
refuses is what the fused loop refuses too (a free hand scale, ortho): there autograd's gradients stand.  Cost of the exact
path: nothing at one clip, -2 % on an 8-clip batch (EXPERIMENTS.md): the sweeps are bound by LDS and dependent loads, not by the
divisions; the hand side's kernels did not change but for the sin / cos.

Teacher-forced lock-step parity at full size (`bench.lockstep_parity`, `tests/test_lockstep_gpu.py`) stays as the per-step
statement for all loss sets: BEFORE every step the fused loop's parameters are loaded into the (faithful, autograd) oracle,
which evaluates that step there - zero flipped samples, object vertices bit-equal, `loss_sil_obj` equal to the last bit, every
loss of the step-1 set within 2.4e-7, gradients within 2e-6 of their largest entry, hand vertices within 6e-5 mm; cfg3's
`loss_collision` up to 1.8e-4 (conditioned on the hand's one-ulp vertices; 1.1e-7 on the HIP vertices); and, new, cfg2 WITH the
ordinal depth term at 30 x 256^2 (losses incl. `loss_depth` within 1e-4, zero flipped samples in the silhouette raster and in
the object's depth render).

## 3. Data layout in HBM (per clip, B frames, S=256 → 512² samples, F faces, V vertices)

| Buffer | Shape / type | Bytes (cfg2) | Producer → consumer |
|---|---|---|---|
| `faces9` (packed face buffer) | (B,F,3,3) f32 NDC | 3.2 MB | `k_setup_faces` → `k_raster_fwd` (record build), sweep work list |
| `boxes` | (B,F) 4×u16 sample box + 2-bit winding mask | 0.7 MB | `k_setup_faces` → `k_raster_fwd`, sweep work list |
| `bin_cnt`, `bin_list` | (B,64) i32 ; (B,64,F) i32 — faces per 64² super-region (128² above 512² samples) | ≤ 23 MB (≈ 0.6 MB used) | `k_setup_faces` → `k_raster_fwd` |
| `idx_map` | (B,512,512) i32, −1 = background | 31.5 MB | `k_raster_fwd` → `k_bwd_lines`, `k_bwd_sweep`, `k_depth_bwd_faces` |
| `alpha16` | (B,32,32,16) u16, 1 bit / sample, tile-blocked | 1 MB | `k_raster_fwd` → `k_bwd_masks` (generic backward only) |
| `pooled`, `dimg` (`gimg`: generic backward) | (B,256,256) f32 | 7.9 MB each | raster → reduce / lines |
| sweep planes | (B,32,32,4,16) u16, **tile-blocked**: the row / column words of both planes of an 8×8-pixel tile side by side = one 128-byte line per tile | 3.9 MB | `k_raster_fwd` epilogue (fused loss) or `k_bwd_masks` → `k_bwd_lines` |
| `lrec` | (4,B,512) lines × 8 × {64 mask bits, sources before them} (16 B) — one cache line per line | 7.9 MB | `k_bwd_lines` → `k_bwd_sweep` |
| `lsum` | (B,2 axes,512 lines) × 2 planes × {first set position, last + 1, 16-bit mask of non-empty 64-sample words} (16 B per line) | 0.5 MB | `k_bwd_lines` → `k_bwd_sweep` stage 1 |
| `srcs` | (4,B,512) lines × ≤512 × {d1, g, owner} | 377 MB reserved, ≈ 0.3 MB touched | `k_bwd_lines` → `k_bwd_sweep` |
| `owned` | (B,2F) u8 | 0.2 MB | `k_raster_fwd` → work list |
| sweep work list: `tab`, `offs`, `ufirst` | active faces × 64 B {corners in pixels, cumulative item counts of the 12 (winding, edge, axis) families, first item} ; first face of every 64-item unit | ≤ 5.8 MB + 0.4 MB + 1.4 MB (≈ 45 % of the faces active) | compaction blocks of `k_bwd_lines` → `k_bwd_sweep` |
| `parts` | (B,F,3,2) f64 corner gradients: exact sums on the 2^-44 grid | 4.3 MB | `k_bwd_sweep` → `k_bwd_gather` / `k_rigid_bwd_x` |
| MANO model `M` | (145,2334) f32 = [posedirs; shapedirsᵀ] | 1.35 MB | constant, L2-resident |
| SDF: `tris` | (B,F_k,16) f32 packed triangle + box | 8.7 MB | `k_sdf_tris` → `k_sdf_dist` |
| SDF: `masks`, `needm`, `need_list`, `phi` | 32² u32 rows ; ≤ 32³ voxel ids / distances per (mesh, frame) | 0.5 MB + sparse | sign → need → distance → sample |

Everything of one clip stays resident (≈ 0.5 GB reserved, ≈ 80 MB touched per iteration); a 288 GB GPU holds hundreds
of clips.

**Clip batches (cfg4 / cfg5).**  C clips of equal shape are ONE set of the arrays above with the frame axis C·B long: clip
c = frames [c·B, (c+1)·B).  Per-frame tensors (poses, vertices, masks, cameras, every workspace array) are simply longer;
per-clip scalars become (C,) arrays (`int_scales_*`, `Σkeep`, scale-prior unit gradients); the loss / metric slots are a
(C, 14) table (`out_stride` = 14 floats), the log is (steps, C, 14); reduce workspaces hold C slices of 576 floats
(partials + ticket) back to back; the sweeps' work list gives every clip its own run of compaction blocks.  `ClipBatch`
(`homan_amd/clipbatch.py`) builds this from C `HOMan` models and re-points each model's Parameters to views of the batched
storage, so every model ends up holding its own result.  8 clips of cfg2: ≈ 4 GB reserved.

## 4. Kernels (all hand-written HIP, wave64, no MFMA — there is no dense contraction on this path)

`bound` = which roofline limits the kernel at its design point; bytes are algorithmic HBM bytes per launch for cfg2
(B=30, S=256, F=3000, V=1502).  One optimisation iteration = 10 launches (cfg2: setup, raster, lines, sweeps, object
gradients, Adam + log row | MANO forward, pair terms, hand gradients, MANO backward) / 20 (cfg3) on two HIP streams for one
clip; a clip batch keeps the terms in launches of their own and adds a third stream (15 / 22 launches).  Every kernel takes a batch of clips in ONE launch: per-frame kernels index their per-clip
scalars by `frame / clip_len`; reduction kernels add a grid row (or ticket) per clip, each clip's sums formed by its own
workgroups in the order of a single-clip launch - which is why a batched step is bit-identical to C single steps.

| Kernel | Work decomposition | Bound | Alg. bytes / launch |
|---|---|---|---|
| `k_setup_faces` | thread / face: optional rigid transform of the mesh-space vertex (the silhouette chain does not wait for `hm_rigid_fwd`), projects its 3 vertices (no separate projection launch), windings, tight sample box; bins the face into 64² super-regions (128² above 512² samples; LDS-aggregated counts, one global atomic per (workgroup, bin)); extra workgroups write the camera-space vertices for the other losses (same arithmetic as `k_rigid_fwd`: the hand-side stream no longer opens with a transform launch) | HBM | B·(V·24 + F·(12+36+8+2+5)) = 6.7 MB |
| **`k_raster_fwd`** | workgroup = one 32×32-sample region: scans the faces of its super-region; candidates are split by WINDING CLASS - the class that holds the camera-facing surface of the mesh (a scheduling hint set by `calibrate()` from the index map; any value is correct) fills the candidate array from the front, the other from the back; one thread per candidate builds the face record in LDS (+ the nearest depth the face can produce) - BOTH classes in one pass, wave 0 up to 64 near-class records while wave 1 builds up to 64 far-class records (one round trip to the packed faces and one stretch of record arithmetic per round instead of one per class: the far class's pass had been a quarter of an active workgroup's time with three waves at its barrier; per-class prefix sums are per-wave scans); the (candidate, 4×4-sample block) units are **flattened** over the 256 threads (binary search of the exclusive unit counts); visibility = `ds_min_u64` on an LDS z-buffer keyed (depth bits ≪ 32 \| face) = strict z test in ascending face order, bit-exact.  The near class runs first; then per 4×4 block the largest owner depth is taken (`hz`), and the units of the far class are first tested against it, 64 per wave trip, the survivors queued per wave and the unit body run on FULL waves of survivors (a divergent early-out would leave the wave paying for its one visible unit: on a closed mesh ~85 % of the far units are hidden).  Sample positions of power-of-two grids by one multiplication (eight IEEE divisions per unit before).  Epilogue per 8×8 output tile: index map, pooled silhouette, fused masked-MSE terms, alpha plane and the four sweep bit planes of the backward (one full cache line per tile, ballots picked with selects: no scratch); in a fixed loop (`persistent_outputs`) an empty bin in front of outputs that already hold the empty pattern leaves before touching LDS (60 % of the workgroups).  The inside test of a unit is arithmetic, not compares: `rv < cv` is the SIGN BIT of `rv - cv` (after `rv + 0.0f`, which turns the one case where the sign lies, -0, into +0), the three edges' differences OR-ed into sixteen accumulators and shifted into the mask by one `v_alignbit` per sample - 156 instead of 221 instructions per unit (round 5); faces with a vertex projected beyond 1e15 are culled by the face setup and by the oracle (their edge functions overflow to inf - inf).  Wave-uniform values (work-order entry, wave index and everything derived) go through `readfirstlane` into scalar registers.  25 KB of LDS and 80 VGPRs (no spills): 6 workgroups per CU (7 per CU at 72 registers: measured slower, EXPERIMENTS r5) | latency × residency (5-6 dependent round trips per active workgroup), VALU-bound only when the GPU is full (clip batch, pose initialisation) | B·(F·49 + 512²·4 + 4·S²·4 + 5·512²/8) = **72.2 MB** |
| `k_setup_faces_multi`, `k_raster_fwd_multi` (`hm_sil_fwd_multi`, round 6) | the two kernels' BODIES (`setup_faces_body`, `raster_fwd_body`: `__forceinline__` functions over the kernels' argument lists) called with one of up to four per-render argument blocks held in the kernarg segment; a workgroup finds its render by a scalar search over the renders' first workgroups (raster) / first frames (setup: grid = largest block count x all frames, surplus blocks leave at once) - every argument stays a scalar load, same registers (80 / 38), no scratch; renders differ in mesh (V, F, vertex and face arrays), cameras, masks, outputs, render size and workspace, each backward runs on its own workspace as before.  Used by the fused loop for the ordinal depth term (silhouette render + the object's depth render: 41 + 7 + 34 µs of launches become 56); it is also the raster-stage entry point for frames of DIFFERENT meshes (one render per mesh) | as the bodies | per render as above |
| `k_sil_reduce` | block / frame + last-block finish.  One clip: the same body rides as B extra workgroups at the front of the `k_bwd_lines` launch (`hm_sil_bwd_clips(..., loss_out)`): the value is only logged, so it costs no launch; a clip batch keeps it on the third stream | latency | 0.5 MB |
| `k_bwd_masks` | generic backward only (arbitrary `dL/dsilhouette`, or a negative loss weight): wave / tile, sweep planes via ballots | HBM | ≈ 21 MB |
| `k_bwd_lines` | one DPP row (16 lanes) per TWO consecutive lines of a (plane, orientation, frame), 32 lines / workgroup (their mask words arrive in the same 4-byte loads; the launch is about one resident round of workgroups): expands a bit line into a position-sorted array of sources {d1, g, owner} + a 16-byte record per 64-bit word {mask, sources before it} (row scan).  Its first ⌈B·F/256⌉ workgroups build the **work list** of the sweeps instead: faces that own a sample → 64-byte records laid end to end in one global item space (block scan, one 64-bit atomic per block; a block's items start on a 64-item boundary so that the composition of every unit — and with it every summation order — is independent of the order in which blocks draw their bases).  Round 5: the 32 lines of a workgroup share plane, orientation and frame - decomposed once per workgroup in scalar registers (two 64-bit divisions per lane before) - and every address is a scalar base + 32-bit byte offset (`hm_at<W32>`) | latency | B·(4·512²/8 + S²·4 + 512² + F·46 [reads] + 4·512·8·16 + 2·512·16 + F·56 [line records, summaries, work list]) = 37.2 MB (+ 12 B per source and orientation, data dependent) |
| **`k_bwd_sweep`** | persistent waves; a **unit** = 256 consecutive (face, winding, edge, axis, d0) items of the global list, whichever faces they belong to (a big face spreads over several waves, small faces share one), handled per pass of ≤ 16 faces staged in LDS together with their per-(face, family) constants - edge slope, first line, end-point order: one IEEE division per family instead of one per item, built by the wave right after the staging - and per-(face, axis) inward ranges.  **Stage 1** (every item, 64 per trip, four trips whose loads are all in flight before the first is tested): family by a 4-step search of the cumulative counts, line geometry from the family constants, then three loads requested together: the line's 16-byte summary, the owner of the sample just inside the edge and the alpha word of the sample just outside - the outward sweep needs a sample this winding owns AND a plane-0 source beyond the edge (exact from first / last position), the inward sweep an empty sample outside AND a plane-1 source inside the triangle's extent (positions + word mask); in the steady state of a fit 71 % of the items stop here (1.95 M items → 557 k, of which 555 k do have pairs) and the rest are queued with the two decisions.  **Stage 2** (queued items on full waves): two 16-byte record loads + popcounts give the slices `[lo, lo+nb)` of the line's source array; the (item, source) pairs are **flattened** over the wave, four consecutive pairs per lane, item of a pair by scatter + max-scan; every term is `diff / dist` with IEEE divisions and is rounded to the 2^-44 grid; per-lane running sums in DOUBLE, flushed into the face's six double LDS accumulators (`ds_add_f64`) when the (face, corner) target changes (behind a full round, rows of 16 lanes that flush one target add up with a DPP tree first and flush once: long sweeps early in a fit put dozens of lanes on the same word); a face inside one unit is stored, a face cut by unit boundaries is added by its units with hardware double atomics onto a zeroed target - the sums are exact, so any order gives the same value, nobody waits, and the per-unit partials + tickets of rounds 2-3 are gone; **XCD-aware**: each XCD (workgroup id mod 8) takes one contiguous eighth of the units, so a frame's index-map lines, line records and source slices are fetched into one L2 instead of eight Addresses (round 5): scalar base + 32-bit byte offset formed in 32-bit arithmetic (`k_bwd_sweep<W32>`, instantiated while the source arrays stay below 4 GB; stage 1 spent 30 of 66 instructions per trip on 64-bit address arithmetic, now 12) | VALU issue while every wave is busy (`valu_frac` 0.68; in the 8-clip batch 0.78), then the dependent-load latency of the units with many pair rounds; WORK-bound, not balance-bound: dealing units out dynamically, cost-sorted orders and a chunk list that spreads long sweeps over all waves were each built, bit-identical, and lost (EXPERIMENTS r4 / r5) | B·(F·(68+48) + 512²·4 + 512²) = **49.8 MB** |
| `k_bwd_gather` | thread / vertex over CSR adjacency: one `double2` per (face, corner), summed exactly, rounded once + projection backward.  Autograd path only: the fused loop gathers inside `k_rigid_bwd` (`hm_rigid_bwd_sil`) | HBM | B·(F·24·2 + V·24) = 5.4 MB |
| `k_rigid_fwd/bwd`, `k_rigid_bwd_x` | forward: thread / vertex; backward: grid (frame, 256-vertex chunk), sums up to four weighted per-vertex gradient terms + a per-frame vector + optionally the silhouette gradient gathered from the sweeps' per-corner output (no gather / linear-combination launches), 13 block sums behind two barriers, per-frame ticket, rot6d backward by the finishing workgroup.  The OBJECT's backward of the fused loops is `k_rigid_bwd_x`: the same work with every dependent load stage (CSR offsets → corner items → per-corner doubles) issued for all of a thread's vertices at once, the 13 sums EXACT (addends on the 2^-44 grid, DPP reductions on doubles, chunk records of doubles: any split of the vertices gives the same floats), and the object's temporal-smoothness gradient formed in the kernel from the camera-space vertices of the neighbouring frames - on the step-1 sets the object's chain waits for nothing the hand-side stream produces | latency (a chain of ~6 round trips: 15 µs for 180 workgroups) | ≤ 4·B·V·12 |
| `k_mano_fwd`, `k_mano_bwd` | forward: (13 vertex chunks × ⌈B/4⌉) blocks, a workgroup = one chunk of 64 vertices for FOUR consecutive frames, wave f owning frame f: four chains prepared side by side, then every wave streams its 37 rows of the blend matrix `M` ONCE and accumulates them for all four frames (the 1.35 MB matrix crosses L2 once per four frames: a 240-frame batch used to pull 324 MB through L2 per launch), wave f finishes frame f (partials, skinning, rigid transform, state); per frame the arithmetic and its order are unchanged, so frame grouping is invisible in the results.  Backward: (13 × B) blocks; kinematic tree staged in LDS, chain level-parallel; `M` rows streamed coalesced with all of a wave's rows requested before the first is reduced; rigid hand transform fused into the forward epilogue; the forward keeps the chain state + posed vertices for the backward; backward = ONE launch: chunk partials, then the frame's last workgroup (per-frame ticket) runs the chain / Rodrigues / PCA backward with the prior folded in; `k_mano_bwd<true>` (`hm_mano_bwd_rigid_clips`, one clip) also does the hand's rigid backward - no mesh-gradient buffer, no `k_rigid_bwd` launch for the hand | L2 / latency | 2·C_mano + 2·B·778·12 |
| `k_hand_terms` | 2-D reprojection + temporal smoothness + priors of the hand in one launch, one ticket | latency | ≈ 1.5 MB |
| **`k_pair_terms`** (`csrc/pairterms.hip`) | one clip: the terms that start from the two vertex buffers and feed nothing to each other as BLOCK RANGES of one grid - [search \| interaction \| hand terms \| object smoothness] - each with its own reduce workspace and ticket; the search is the metric-only one on the step-1 sets and the FULL one (`nn_full_body`, the body of `k_nn`, with the contact launches behind it) when the contact term is on (cfg3: 214 → 202 µs per iteration).  The bodies are shared with the stand-alone kernels (`pair_bodies.h`: device functions on virtual block coordinates), so the floats are the same either way (`test_pair_terms_launch_equals_its_four_entry_points`, and every batched == single test: batches use the stand-alone launches).  Four launches and three graph edges of the hand-side chain become one: 5100 → 5430 it/s | latency | ≈ 2 MB |
| `k_smooth`, `k_inter`, `k_contact_hand` | grid-stride + "last block finishes" ticket | latency | 10–100 KB |
| `k_contact_both` | contact term, both sides in one launch (object meshes of ≤ 4096 vertices; larger: `k_contact_hand` + `k_contact_obj` over ranges of 4096): per frame the hand vertices' gradients go to memory and, from the register, into 64-bit fixed-point LDS accumulators of the object's vertices (order-free integer sums: deterministic) | latency | B·(778·36 + V·12) |
| `k_nn` | (contact term on) 128 hand vertices (2 / lane) × 4 waves splitting the object vertices; 64-vertex groups broadcast with `v_readlane` (SGPR operands) | VALU | B·(778+V)·12 |
| `k_nn_min` | (step-1 sets: the search only feeds the logged hand-object distance) bounding spheres of 64-vertex groups of the Morton-ordered rigid mesh, carried from a mesh-space table into the frame; per group a lower bound for the workgroup's 128 hand vertices (nearest centre distance − radius); the four groups with the smallest bounds are scanned first, one per wave, and their exact minimum is the upper bound - with "centre distance + radius" 21.5 of the 24 groups of the bottle passed when the hand touches it, now 5.6 do; scans run four object vertices per trip on scalar trip counts with the minima kept as integer bits (a wave is alone on its SIMD: the independent chains hide the arithmetic latency); the result is the same float as `k_nn`'s.  Round 6: **seeded** - `nn_seed` (caller-owned, carried from launch to launch) holds per frame the vertex pair that held the minimum at the last launch; that pair's distance NOW is an upper bound known before anything is scanned, the four unconditional first scans go, and a workgroup scans only groups that can beat it (most: none); the clip's finishing workgroup turns each frame's winner (vertex and group, tracked per lane at group granularity) into the next seed by one 64-lane step.  Any seed content is valid - a pair is a pair -: the result is exact whatever it holds  42 → 24 µs stand-alone, and this launch range was the longest link of the hand-side chain | latency | B·(778+V)·12 |
| `k_contact_obj` | block / frame: hand-vertex gradients added into an LDS accumulator with 64-bit **fixed-point** atomics (order-independent ⇒ deterministic without a sort; the picks are skewed onto a few object vertices) | LDS | B·(778·16 + V·12) |
| `k_sdf_boxes`, `k_sdf_tris`, `k_sdf_need`, `k_sdf_dist`, `k_sdf_sample` | AABB + normalise; thread / triangle: packed record + +x ray parity of the few (y,z) rows under the triangle (`atomicXor` of 32-bit inside masks); thread / sample: marks touched inside voxels, first setter appends to the grid's list; workgroup / listed voxel: nearest-vertex seed + box-pruned scan of the packed triangles (a single wave was ~70 dependent round trips for one voxel); thread / sample: trilinear value + gradient, ticket reduction | latency | ≈ B·(778+V)·36 + 8.7 MB |
| `k_depth_bwd_faces`, `k_depth_bwd_gather` (a19) | a wave per run of 4 (frame, face) slots: a lane per (slot, winding) finds the windings that own a sample in one coalesced trip (idle ones get their zeros there), then the wave walks each live winding: strides the face's sample box, tests ownership in the index map, reduces the three sums `A_k = Σ g·zp²·w_k` the NMR depth backward factors through (DPP); thread / vertex gather + projection backward.  Round 6: **sparse** (`hm_depth_bwd_sparse`) - the ordinal term's gradient images are zero wherever render and annotation agree on the order; `k_ordinal_depth_bwd` leaves one byte per (frame, pixel row, 64-pixel segment), a winding whose sample box touches no flagged segment gets its exact zeros in the prologue, a frame without flags its zero vertex gradients without the gather's two round trips: 36 + 31 → 11 + 14 µs and 16 + 16 → 13 + 7 µs at cfg2 | HBM (index-map reads) | B·(512²·4·ρ + S²·4 + F·(44+72)) (ρ ≈ box overlap) |
| `k_ordinal_depth`, `k_ordinal_depth_bwd` (a19) | 16 chunk workgroups per frame; a frame's record collects them with 64-bit INTEGER atomics (pixel counts packed, softplus sums in 2⁻³² fixed point: order-independent, so deterministic), the chunk that completes a frame learns it from the record atomic it issues anyway (arrival count in the record's spare bits) and only that one draws the clip's ticket; seven block sums behind two barriers; last workgroup finishes the clip; element-wise backward.  In the fused loop the object's depth render and depth backward ride the calling stream (behind the silhouette raster / behind the sweeps), the hand's the side stream | HBM | B·S²·(4·4+2) fwd, + B·S²·8 bwd |
| `k_adam` | one launch for all tensors (pointer table), bias corrections in double by square-and-multiply (a function of (beta, t) alone: the oracle's Adam forms the same doubles), zeroes grads; the last workgroup (ticket) bumps the device step counter.  One clip: an extra grid row writes the log row of the step being taken (weighted total + every loss / metric slot) before it draws its tickets (`hm_adam_step_log` = `hm_log_total_clips` + `hm_adam_step`, same floats, one launch less on the tail) | latency | 28·79·B |

Whole iteration (SURVEY §8d byte model): 144.1 MB (cfg2), 161.8 MB (cfg3).

CDNA4 specifics used: wave64 ballots for binning / compaction; **flattening of heavy-tailed per-item work over the
wave or workgroup** (DPP `row_shr` / `row_bcast` inclusive scans, then an LDS binary search or a scatter + DPP max-scan
to map work back to its owner) in both heavy kernels, at two levels in the sweep (items over the waves of the whole grid,
pairs over the lanes of a wave) — faces are ~50 samples and sweep items ~5 sources on average but hundreds at the tail,
lane-serial loops left > 90 % of the lanes idle and a wave per face left the grid waiting for its largest face;
commutativity instead of ordering (a value cut in two is added with two hardware float atomics onto zero: 0 + a + b is
the same float in either order); `ds_min_u64` z-buffer in LDS; `v_readlane` → SGPR broadcast; DPP wave reductions (always at wave-uniform points:
a lane that has left a loop feeds them stale registers); single-launch grid reductions whose partial records travel as
agent-scope (sc1) stores / loads instead of release / acquire fences (a release fence writes back the whole L2 of the
XCD — expensive next to a kernel that is filling it); per-bin tickets instead of one ticket word for 7680 workgroups
(same-address returning atomics serialise: +43 µs measured); order-independent integer / XOR / min atomics wherever a
scatter would otherwise need a sort to stay deterministic; `s_setprio` on the latency-bound kernels (a lone young wave
next to four issue-bound persistent sweep waves got ~1/5 of the SIMD's issue slots: the hand-side kernels ran 2–5× longer
whenever they overlapped the sweep); hipGraph replay of the whole iteration (forward + backward +
Adam + logging, no host sync) on three captured streams.

Stream structure of the fused loop (`FusedStepper`): A (face setup incl. the object's rigid transform and the camera-space
vertices → raster → lines [+ loss reduction] → sweeps → object pose gradients incl. the silhouette gather), B (MANO + hand
transform → pair terms → hand + MANO backward); B forks off A right after the face setup (`hm_sil_fwd_phase_clips`); join →
Adam + log row.  A clip batch adds C (silhouette reduction + log rows), lets B's pair-wise terms wait for the END of the line
expansion (`hm_sil_bwd_phase_clips`: that kernel is latency-bound and takes 200 instead of 150 µs next to neighbours that hold
its wave slots) and runs 1024 instead of 1280 persistent sweep workgroups so that the hand's gradient launches find registers
next to them.  HIP-runtime and hardware facts that shaped it (all measured, EXPERIMENTS.md):
* stream capture crashes at replay when a forked stream rejoins a stream other than the one it forked from (a third branch
  for search + contact next to the collision term; the collision chain on the third stream joining the side stream), and
  with the depth term's launches on the side stream a three-stream graph dies too (depth: two streams); torch hands streams
  out of a pool of 32, so all steppers share one verified set of distinct streams (`_loop_streams`);
* consecutive kernels of one queue follow each other without a gap; every cross-queue edge costs 5-10 µs.  At one clip the
  silhouette chain IS the iteration; the hand side is hidden under it (removing the MANO backward altogether: +1.5 %);
* kernels that share the GPU stretch each other: six rasteriser workgroups fill a CU's LDS and registers, five sweep waves
  per SIMD leave 32 registers - whether a hand-side kernel runs under a heavy kernel or after it is a matter of residency,
  which the LDS ballast knobs (`hm_tune_*`) and the sweep's workgroup count steer per loss set.

## 5. Measured (MI355X, round 6; evidence under `profiles/r06_*`, regenerated by `tools/profile_round.sh r06`)

`python bench.py` prints ONE compact JSON line (<= 2 KB: the contract's keys, `roofline`, `cpu_baseline`, one number each for
the `steady_state`, `multi_clip`, `cfg2_depth` and `cfg3` legs) and writes the full record - per-kernel tables, notes, with `--parity` the HIP-vs-oracle
legs of `bench_parity.py` - to `gpurun_out/bench_detail.json` and stderr; the default run takes ~25 s on the GPU box.  (Round 4's
line was 23 KB, parity traces included, and came back from the driver as `parsed: null`.)  Default workload: cfg2, 400 steps
after 20 warm-up, then - same process, same fit, same hipGraph - a `steady_state` leg (iteration >= 400, 1000 timed iterations)
so that ONE line carries both regimes whatever `--steps / --warmup` the driver passes (it passes `--steps 20 --warmup 5`:
iterations 5-25 of a fresh fit).  `profiles/r06_bench_*.json` are the full records, `*_line.json` the printed lines.

| Quantity | Value |
|---|---|
| cfg2 (1 clip, 30 frames 256², bottle 3000 faces, step-1 losses), headline | **@CFG2@ it/s** (@CFG2MS@ ms / iteration); `steady_state` **@STEADY@ it/s** |
| the same with the driver's round-1 flags `--steps 20 --warmup 5` (iterations 5-25 of a fresh fit: 5-10 M sweep pairs instead of 1.4 M) | @DRV@ it/s headline, @DRVSTEADY@ it/s in its `steady_state` leg |
| cfg2 as BASELINE.json words it, with the ordinal depth term (`--depth`) | @DEPTH@ it/s |
| cfg3 (step-2: + collision + contact) | **@CFG3@ it/s** |
| cfg4 in miniature: the 8-clip shard of a GPU through `ShardStepper` (`multi_clip`): two clip batches of four side by side since round 6 (one batch of eight: 9 565 it/s) | **@MULTI@ it/s** summed = @MULTIX@ × one clip; whole-iteration roofline fraction @MULTIF@ |
| a heterogeneous shard: 8 clips of 4 shapes (bottle / cube, 30 / 20 frames) through `ShardStepper`, the shape groups' graphs replayed concurrently | @MIXED@ it/s summed (one after the other: 8 330) |
| END TO END (`end_to_end` of the bench line): 16 cfg2 clips x 400 steps through `ClipFitter`, 8 per batch, wall clock from the per-frame input dicts on the host to the results on the host - the first batch builds model / workspaces / graph, the second is copied into the resident stepper | **@E2E@ clips/s** over all 16; a clip of a RESIDENT shape: **@E2ERES@ clips/s**, input load = @E2ESETUP@ of its fit; split: @E2ESPLIT@ |
| cfg5 on one rank (8 clips, step-2, one tied scale, RCCL call issued) | @CFG5@ it/s |
| 2 ranks on ONE GPU through gloo (the driver's `torch.distributed.run` line; weak scaling has nothing to scale on one GPU - this is the N > 1 code path, not a speed-up) | cfg2: @G2@ it/s summed; cfg5 (2 x 4 clips, tied scale): @G5@ it/s, replicas identical |
| object-pose initialisation (SURVEY §8f rank 1): 500 candidate poses of the bottle against one 256² mask, `python bench.py --pose-init 500` | **@POSE@ pose-steps/s** (@POSEMS@ ms per step of 500 poses; a 50-step fit in @POSEFIT@ s); by loop: @POSELOOPS@; CPU oracle @POSECPU@ pose-steps/s |
| free-running parity, cfg2 at full size, 400 steps, HIP loop vs the oracle's reproducible loop (`profiles/r06_freerun_cfg2_400.json`) | every parameter (object pose, hand pose, MANO) bit-equal after every step: @FREEALL@ (object alone: @FREEEQ@), final vertices object @FREEVO@ mm / hand @FREEVH@ mm, largest relative loss difference at any step @FREELOSS@ (section 2; the control `r04_control_cfg2_400.json`: the CPU loop against itself from inputs 1e-7 m apart ends 0.85 mm apart) |
| CPU baseline (oracle loop, 64 host threads) | @CPU@ it/s with the reference's per-step `.item()` logging, @CPUOFF@ it/s without → GPU / CPU ≈ @RATIO@ × (target ≥ 50 ×) |
| whole iteration vs the SURVEY §8d byte model (144.1 MB) | @WHOLE@ of 8 TB/s at one clip |

**Roofline of the heavy kernels, inside the replayed graph** (ROCm allows no timing events inside graphs, so every
workgroup of the three kernels stores `s_memrealtime` at entry and exit, `hm_sil_timestamps`; `bench.py` replays THE SAME
graph 50 more times and averages; `rocprofv3 --kernel-trace --stats` of the same command, `profiles/r06_p_cfg2_headline_kernel_stats.txt`,
agrees to a few per cent, see the note in EXPERIMENTS.md on what the profiler itself moves):

| kernel (cfg2, steady state) | µs / launch | algorithmic MB | GB/s | frac of 8 TB/s | PMC traffic MB | VALU wave-instr | valu_frac (measured mix / flat 4 cycles) |
|---|---|---|---|---|---|---|---|
| `k_raster_fwd` | @RAS@ | 72.2 | @RASG@ | @RASF@ | @RAST@ | @RASV@ | @RASVF@ |
| `k_bwd_sweep` (`roofline.kernel`: the longest launch) | @SWP@ | 49.8 | @SWPG@ | @SWPF@ | @SWPT@ | @SWPV@ | @SWPVF@ |
| `k_bwd_lines` | @LIN@ | 37.2 | @LING@ | @LINF@ | @LINT@ | @LINV@ | @LINVF@ |

The per-launch models above count the kernels' own intermediates (line records, work list) and sum to 159.2 MB; the STRICT
reading of SURVEY §8(d) - its raster stage, 134.7 MB, over the three kernels' summed durations - is the bench line's
`roofline.stage_frac_8d` (@STAGEF@).

`valu_frac` = `SQ_INSTS_VALU` × c ÷ (launch time × 1024 SIMDs × 2.4 GHz), c = cycles per wave64 instruction.  Round 5 left c
open between the counters' 4 (`SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU` = 1.02 quad-cycles) and the guide's 2; round 6 MEASURED it
(`tools/valu_ceiling.hip`, `profiles/r06_valu_ceiling.json`: independent instructions at 1 / 2 / 4 / 8 waves per SIMD, all 1024
SIMDs, wall clock at the nominal 2.4 GHz): **two classes** - ~2.3 cycles for `v_add / v_sub / v_mul / v_fma_f32`, `v_add / v_sub_u32`,
`v_mov`, `v_and / v_or / v_xor` (the guide's number), ~4.2 for everything else this code is made of: `v_cmp`, `v_cndmask`,
shifts, `v_min / v_max`, conversions, every three-operand integer op, DPP, doubles, packed fp32 (the counters' number);
`v_rcp_f32` 8.1, the IEEE division 47.9 per quotient; ONE wave alone on a SIMD issues an instruction per ~5 cycles whatever the
class.  Priced with the kernels' static opcode mix (`tools/valu_mix.py`, `profiles/r06_valu_mix.json`) c = 3.3 (raster), 3.5
(sweep, lines): the table's first number; the second is round 5's flat 4.  Reading: traffic is AT the byte model (the kernels are
not HBM-bound); at one clip they issue VALU on a half to two thirds of all SIMD cycles, the rest is the dependent chain of every
workgroup (below); in the 8-clip batch the sweep reaches ~0.7 - close to, not at, its issue ceiling - and what remains there is
the mix itself (a quarter of the sweep's instructions are selects and compares at 4 cycles).

**Ledger of the iteration** (rocprofv3 per-kernel averages, µs; `tools/ledger.sh`, `profiles/r06_ledger_*`; b1 = ONE frame of the
cfg2 clip, every kernel at its latency floor):

| kernel | b1 (1 frame) | cfg1 (10 × 128²) | cfg2 (30 × 256²) | dependent global round trips of its critical workgroup |
|---|---|---|---|---|
| `k_setup_faces` | 6.5 | 6.3 | 9.2 | kernargs, faces → vertices, bin count (returning atomic), list store |
| `k_raster_fwd` | 18.2 | 17.3 | 49.6 (40 in-graph) | kernargs, launch-order entry, bin count + state, list, boxes, faces, ticket, keep / ref |
| `k_bwd_lines` | 9.7 | 10.9 | 21.1 | kernargs, plane words, gradient + owner gathers per word (work-list blocks: scan + one atomic) |
| `k_bwd_sweep` | 17.5 | 18.8 | 46.1 | totals, unit table, face records, summaries + owner + alpha, line records, sources |
| `k_rigid_bwd_x` | 15.5 | 9.0 | 17.3 | offsets → items → corner sums, chunk records + ticket, records |
| `k_adam` | 4.1 | 4.3 | 4.5 | slots, tensors, ticket |
| hand side: `k_mano_fwd` / `k_pair_terms` (cfg1, b1: `k_v2d`) / `k_mano_bwd` | 13.3 / 5.0 / 30.8 | 13.7 / 5.4 / 32.0 | 16.6 / 30.3 / 42.2 | |
| iteration, un-profiled | 77.8 | 77.2 | 154.0 (150.2 with this round's launch graph) | |

One frame costs half of thirty frames: the iteration is the sum of its kernels' own latency chains.  What such a chain is made of,
from per-workgroup phase stamps of the rasteriser (debug build `-DRASTER_TRACE`, `tools/raster_trace.py`,
`profiles/r06_raster_trace.txt`): an ACTIVE raster workgroup lives 11.9 µs alone on its CU (scan 3.2, records 1.5, its near units
3.6, barrier + hidden-block depths 0.3, far units 1.1, epilogue 2.2) and 16.6 µs in the 30-frame launch (slowest tenth 23 µs);
2 746 of the 7 680 workgroups are active, six fit a CU, so the launch is TWO ROUNDS of them (start offsets: median 2 µs, 90th
percentile 21.8 µs) = 41 µs.  No phase is more than 30 % of a workgroup; the two attempts to shorten one (scan as one round trip
with box-carrying bin entries requested speculatively; two covered samples per trip of the unit body) measured +2 µs and ±0
(EXPERIMENTS.md).

Pose initialisation (`bench.py --pose-init 500`, its own line with `roofline`; `profiles/r06_p_poseinit_kernel_stats.txt`,
`r06_pmc_poseinit.json`): per step of 500 candidate poses `k_bwd_sweep` @PISWP@ µs (algorithmic 338 MB → @PISWPG@ GB/s =
**@PISWPF@** of HBM peak, the dominant kernel), `k_raster_fwd` @PIRAS@ µs (@PIRASF@), `k_bwd_lines` @PILIN@ µs (@PILINF@); all
throughput-bound at 500 frames per launch (these are the launches of ONE loop over all 500 candidates, alone on the GPU - the shape
the PMC passes were taken on).  The resident fitter (`PoseFitter`) walks the candidates of a fit as THREE independent loops - own
rasteriser context, Adam state and hipGraph each - replayed side by side on streams of their own: a step is one dependent chain
whose small launches and tails are a seventh of it, and with three chains in flight they run under the other groups' raster and
sweeps: 476 k -> 562 k pose-steps/s, bit-identical (the one cross-candidate rule, the best-ever bookkeeping of reference
`pose_optimization.py:340-353`, is applied after the last step over per-step records the loops leave: `hm_pose_keep_best_log`).

**What bounds the iteration** (EXPERIMENTS.md, rounds 4-6).  The silhouette chain is serial and it IS the iteration: setup 8.5 +
raster 40 + lines 22 + sweeps 44 + object gradients 17 + Adam 4 µs = 136 µs of kernels + ~14 µs between them = 150 µs; the hand
side (MANO 16, pair terms 28, hand gradients 42 µs) runs under it.  Round 6 split the 20 µs between that and the chain's own floor
(`tools/chain_only.py`, `profiles/r06_chain_only.txt`, one process, steppers built one after the other):
* the silhouette chain ALONE on one queue - no side stream, no fork, no join, Adam over the object's pose only (a floor, not a
  fit) - runs **134.4 µs** per iteration; the shipped graph of round 5 154.1;
* both chains with NO edge between them inside the four-iteration graph (the side stream reads whatever pose it finds - not a fit
  either): 148.2 µs.  So ~6 µs are cross-queue edges and ~14 µs are the two chains sharing the GPU (in-graph stamps raster / lines /
  sweeps 40.3 / 22.4 / 46.4 µs next to the hand side, 38.3 / 17.3 / 43.2 alone): a fully decoupled launch graph (object Adam on the
  calling stream, hand Adam + log row on the side stream, ring buffers for the pose, flags instead of edges) could return 3.8 %
  at most and was not built;
* what WAS built: the side stream forms its own copy of the object's camera-space vertices (`hm_rigid_fwd_clips`, the face setup's
  arithmetic: the same floats) instead of forking off the face setup - a node with a successor on another queue costs its OWN queue
  2-3 µs before its next kernel starts: 154.1 → 151.4 µs; and the metric-only search seeded with last iteration's closest pair
  (the pair-terms launch shorter, the sweeps next to it 46.6 → 43.8 µs): → 150.2 µs.  Bit-identical both.
Earlier findings that stand (rounds 4-5, same-box A/B each):
* **The launch floor is not launch latency.**  A dependent kernel boundary inside a stream costs ~1.5 µs, the turnaround between
  two graph replays ~5 µs (taken out by replaying FOUR iterations per graph).  The floor of cfg1 (77 µs) is the sum of six
  kernels' own chains, which a megakernel keeps; what it would add on this GPU is the hand-off: the XCDs' L2s are not coherent
  with each other, so bulk data passed between phases INSIDE a launch (index map, line records, source arrays: 30-70 MB) must be
  written through or fenced per producer (guide: 1.7-6.5 µs per release with freshly dirtied lines) - what a kernel boundary does
  once, wholesale, in 2-4 µs; grid barriers on 1280 workgroups cost 10-14 µs each.  Not built.
* **One clip is a zero-sum game between the two streams' kernels.**  Raster at 7 instead of 6 workgroups per CU: raster 40 → 43.6,
  lines 25 → 19.6, sweeps 45.5 → 48 µs, iteration -1.2 %.
* **The sweep is work-bound, not balance-bound**: dynamic unit hand-out (one queue head per XCD) 51 → 60 µs; heavy sweeps through
  a chunk list and a second launch ±0; long sweeps walked by their whole wave -2 %.  Its LDS conflicts cost nothing in the steady
  state (the LDS pipe is busy 29 % of the launch) and ~5 % at iteration 0, where rows of 16 lanes that flush one key now combine
  with a DPP tree first.
* **Instructions pay where the GPU is full**: 32-bit byte offsets off scalar bases, the scalar line decomposition, the sign-bit
  inside test: cfg2 steady +2.0 %, 8-clip batch +2.5 %, iterations 5-25 +2.9 %, pose initialisation +3.1 %.
Cumulative against round 5 (its numbers in brackets): steady @STEADY@ (6 515), 8-clip batch @MULTI@ (9 379), the driver's flags
@DRV@ (5 153), cfg3 @CFG3@ (5 457), cfg2 WITH the depth term @DEPTH@ (3 716), pose initialisation @POSE@ (479 199).  VERDICT r5's
targets: cfg2 + depth >= 4 300 - @DEPTH@ (met on the headline window, iterations 20-420 of a fit; 4 130-4 160 over iterations
400-700, the leg of the default line); pose initialisation >= 550 k - @POSE@ (met: the candidates as three independent loops);
the 8 clips of a GPU >= 10 000 - @MULTI@ (met: as two batches side by side); cfg2 steady >= 7 000, cfg1 floor <= 62 µs: NOT met - the chain alone on one queue would run 7 440 it/s (cfg1: 63.9 µs), the two chains without any edge 6 750; what is
between those numbers and the shipped graph is the hand side sharing the GPU, and the kernels' own chains were not shortened
(two structural attempts on the raster measured ±0 / +2 µs).

## 6. Multi-GPU

The path shards by clip with no data-path collective (cfg4).  One process per GPU; `dist.shard_clips` deals contiguous,
balanced blocks of clips to the ranks (64 clips / 8 GPUs = 8 each; 9 / 8 = 2,1,1,...), and each rank optimises ITS clips
through `dist.optimize_clip_shard(models, ...)` → `ShardStepper`: clips that agree in shape (frames, object topology, hands,
sizes) form ONE clip batch - every kernel launched once per iteration over all of them, one Adam state per clip - and the
batches of different shapes (a real Core50 shard: every clip its own object mesh, `homan/datasets/core50.py:22-42`, and its
own length, `fit_vid_dataset.py:190`) replay their own hipGraphs CONCURRENTLY, each on a stream of its own (independent clips:
no ordering between groups at all without a tied scale; with one, the two halves of every group's iteration run side by side
around the rank's single all-reduce).  A one-clip graph is ~90 µs of launch and edge latency around ~70 µs of kernels, so
groups side by side fill what a sequence leaves idle: 4 shapes × 2 clips 8 330 → 9 750 it/s, eight one-clip groups 7 800 it/s
(8 clips of ONE shape as a batch: 8 900-9 200).  Configurations the fused loop takes one clip at a time (two hands per frame,
`inter_type="min"`) become singleton groups of the same mechanism.  Every clip ends up with exactly the result of optimising
it alone, bit for bit (`tests/test_clip_batch_gpu.py`).  What is NOT built: one launch per kernel over clips of DIFFERENT
shapes (per-clip CSR offsets in every `hm_*_clips` kernel).  How much it matters for the reference's own driver: a run of
`fit_vid_dataset.py` cuts every sample to the same `--frame_nb` frames (`fit_vid_dataset.py:50,180`) and Core50's model table holds two dozen object
meshes (`homan/datasets/core50constants.py:18-130`, loaded by `core50.py:18-45`), so the clips of a rank's shard fall into a handful of (mesh, length)
groups and each group IS one clip batch; what is left heterogeneous replays concurrently (8 clips of 4 shapes @MIXED@ it/s against
@MULTI@ for 8 clips of one shape).

A process that walks a dataset does it through `ClipFitter` (`jointopt.py`): per shape signature ONE resident stepper -
device buffers, workspaces, ONE captured hipGraph - into which the next clip of that shape is copied (`FusedStepper.reload`,
`HOMan.load_clip`, `hm_sil_invalidate_outputs`; 2.8 ms per clip = 6 % of a 400-step fit) instead of building model,
~0.5 GB of zero-filled workspace, calibration and capture again (about twice a fit); least-recently-used shapes are evicted,
the number of live graphs is bounded by the number of resident shapes (`tools/soak_dataset.py`: 40 clips of 4 shapes, 4
graphs, no memory growth).  Results are bit-identical to fresh fits.

cfg5 ties ONE object scale across all clips of all ranks (an extension; the reference's scale is per clip,
`homan/homan.py:121-130`).  The tied loss is the sum of the clips' losses, so the scalar's gradient is the sum of the clips'
gradients (each with its own scale prior).  Every clip keeps a replica; per step: per-clip gradients, the sum over the rank's
clips, ONE all-reduce (sum) of 1 fp32 over RCCL/xGMI issued on the compute stream between the two captured halves of the
iteration (no host synchronisation), the global sum written to every replica, identical Adam steps - replicas stay
bit-identical.  `ShardStepper` issues exactly one broadcast at the start and one all-reduce per step whatever the rank's
number of shape groups (also zero: a rank without clips), so uneven shards cannot desynchronise the ranks.  4-byte message ⇒
latency-bound; link bandwidth and ring-vs-tree are irrelevant.  Where the tied scalar GOES is the objective's doing: in the
cfg5 bench run it grows 1 → 2.65 over 400 steps - with the object's depth free, scale and depth trade off along a valley of
the silhouette term, the contact term (mean tanh of the distance from every hand vertex to the NEAREST object vertex) falls
when the object's surface comes closer to the hand, the prior's weight is 1e-3, and Adam moves a parameter whose gradient
keeps its sign by ~lr per step; the CPU oracle driven through the same tied loop takes the same path
(`test_tied_scale_trajectory_matches_the_cpu_oracle_loop`).

**N > 1 on hardware.**  No multi-GPU node is available to this build; the scaling curve is the driver's to measure.  What IS
exercised on the one GPU: 2 and 3 processes sharing `cuda:0` under `gloo` (device tensors staged through the host there,
`dist._staged`; under nccl = RCCL the collective runs on the device tensor) run `dist.optimize_clip_shard` - the fused
launch sequence, two hipGraphs around the collective - on even, uneven ([2,1]) and empty ([1,1,0]) shards: replicas
bit-identical across ranks, 2 ranks × 1 clip bit-identical to one 2-clip batch, un-tied shards bit-identical to solo runs
(`tests/test_dist_gpu.py`); cfg5 itself at full size (8 × 30 × 256², step-2, tied scale, one-rank nccl group);
`bench.py --gpus 2` lines under `profiles/r03_bench_*_gpus2_gloo.json`.

### Two hands, left hands (`hand_nb = 2`, `hand_sides = ["right", "left"]`)

`homan/homan.py:62-63,341-358`: hands arrive interleaved frame-major `[h0_t0, h1_t0, h0_t1, ...]`; hand `i` is the strided
slice `i::hand_nb` of every MANO parameter, evaluated by the MANO layer of ITS side and re-interleaved.
* left MANO (`manomodel.py:124-140`): the left model's PCA basis with the y / z components of every joint's axis-angle
  negated before the mean pose is added - folded into the left `ManoContext`'s basis, so one kernel serves both sides.
  `MANO_LEFT.pkl` is read when present; next to the synthetic right hand the left model is its mirror image.
* fused loop: the MANO launches walk their hand's rows in place (`hm_mano_*_rows`), the hand-only terms and the rigid
  backward run once over all rows (the kernels' `hand_nb`), the pair-wise terms see each hand as a dense copy: contact = mean
  over the hands (`lossutils.py:116-127`), interaction = their sum, logged distance = largest per-frame distance to the
  NEAREST hand (`losses.py:207-241`), collision (`lossutils.py:51-59`) = the scene `[hand 0, hand 1, object]` with both hands'
  closed topology in reversed winding, as three two-mesh launches (every mesh's SDF depends on its own vertices only and the
  loss is a sum over ordered pairs, `scenesdf.py:131-146`).
Parity: `tests/golden/ref_step{1,2}_twohands_cube_b4_s64.npz` and `ref_step1_lefthand_cube_b4_s64.npz` come from the
reference's own `HOMan` / `jointopt` at `hand_nb = 2` and with a lone left hand; oracle, HIP model and fused loop are pinned
against them like against the single-hand goldens.

## 7. Out of scope (and why)

Evidence extraction (detectron2 / FrankMocap networks), datasets, tracking, dataset-level evaluation (chamfer / ADD-S on
ground truth, codalab dumps), html / video export: SURVEY §8 marks them out of the hot path and their inputs (weights,
data) are not available.  `contact_mode≠dist_tanh`: non-default branches of a file the reference never reaches.  `hand_proj_mode="ortho"`
(`homan/homan.py:364-371` -> `utils/camera.py:59-105`, non-default) IS built since round 5 - `HOMan.get_verts_hand` places the
hand by its scaled-orthographic camera `cams_hand` (identity rotation, translation from the camera, `s (v + t)` on the rigid
kernels, a twin that detaches the mesh only), `cams_hand` receives its gradient, `optimize_hand_object` takes the graph loop for
it - but as a PARITY-UNPINNED mode: the camera conversion is `libyana.camutils.camconvs.batch_weakcam2persptrans`, a
third-party function that is neither in `/root/reference` nor pinned anywhere, restated here from the camera model (first-order
identity between weak and pinhole camera: `T_z = f / s`, `T_xy = (t - c) T_z / f`), not from its source.  Tests: the identity's
known answers and its inverse on the CPU, HIP == oracle on the GPU (losses 1e-4, vertices 1e-6 m, every gradient incl. the camera's
2e-4 of scale, the first steps of a fit) - `tests/test_ortho.py`.  More than two hands: the reference's own collision term de-interleaves with a stride of 2
(`lossutils.py:58`).  `assign_human_masks` (`homan.py:239-296`): never called by the reference.

## 8. Known gaps / next (ranked)

1. Throughput: steady @STEADY@ it/s (VERDICT r5's target 7 000), the 8 clips of a GPU @MULTI@ (10 000: met by running them as two
   clip batches side by side - one batch of eight stays at 9 565), `k_bwd_sweep` still the longest
   launch at 0.14 of HBM peak - a yardstick it will never approach: at one clip every heavy kernel is the latency chain of its
   workgroups (section 5: one FRAME costs 78 µs, thirty cost 150), in a batch the sweeps issue VALU on ~0.7 of the SIMD cycles
   their opcode mix allows.  Round 6's accounting: the silhouette chain alone on one queue would run 7 440 it/s, both chains
   without any edge 6 750, the shipped graph 6 650 - so the launch graph has ~1.5 % left, the hand side's presence on the GPU costs
   ~9 %, and everything else is inside the kernels.  Open, each worth 1-3 %: selects and compares (a quarter of the sweep's
   instructions, 4 cycles each) turned into adds / ands (2.3); the family search of stage 1 as one ballot per trip; a 16-bit
   index map; the MANO backward's floor (31 µs for ONE frame: its presence is what the lines / sweeps pay 8 µs for).
   Closed with measurements (EXPERIMENTS r6): fewer round trips in the raster's scan (+2 µs), two covered samples per trip of
   its unit body (±0), kernarg preloading (-0.3 µs), a decoupled launch graph (3.8 % at most).
2. One launch per kernel over clips of DIFFERENT shapes (per-clip vertex / face offsets in every `hm_*_clips` kernel) is not
   built.  What stands in for it: `ShardStepper` replays the shape groups' hipGraphs concurrently - 8 clips of 4 shapes @MIXED@
   it/s against @MULTI@ for 8 clips of one shape as two batches side by side (9 565 as ONE batch, the figure VERDICT r5 compared
   with; `profiles/r06_bench_mixed_shard.json`), bit-identical to solo fits.
   Padding clips to a common shape is not an option: padded vertices change the smoothness / interaction normalisers and can
   win the nearest-vertex search.  Round 6 built the raster stage's half: `hm_sil_fwd_multi` launches the face setup and the
   forward raster ONCE over up to four renders of different (V, F) (bit-equal to separate calls,
   `tests/test_raster_gpu.py::test_multi_render_launch_equals_separate_calls`); the line expansion, the sweeps and the loss /
   gradient kernels still take one mesh per launch, so `ShardStepper` keeps its concurrent shape-group graphs - at the rate of ONE
   same-shape batch of eight (VERDICT r5's criterion for this item), below the two-batch rate this round found for one-shape shards.
   The same measurement says what one launch over all shapes would NOT buy: the fewer concurrent batches, the fewer latency-bound
   launches - but one batch of eight is slower than two of four.
3. The pose initialisation at @POSE@ pose-steps/s (VERDICT r5's target 550 k; 476 k before the candidates were walked as three
   independent loops side by side, section 5): sweep and raster throughput-bound at 500 frames per launch (220 M and 161 M VALU
   wave-instructions: `profiles/r06_pmc_poseinit.json`); nothing this round shortened the kernels themselves.  Its line
   expansion's PMC traffic - 532 MB per launch (2 x FETCH 145 MB + WRITE 242 MB) against a byte model of 239 MB + sources - is now
   EXPLAINED, by calibration (`tools/fetch_calib.hip`, `profiles/r06_fetch_calib.json`: kernels of known requested bytes under the
   same two counters): coalesced reads report exactly HALF their bytes at 16 AND at 4 bytes per lane (the guide's x 2 holds for
   both), coalesced stores report their bytes exactly - but an isolated 4-byte gather reports 64 B (128 B after the x 2), an
   isolated 12-byte gather 68 B, 4-byte gathers inside a 4 KB window 28 B, an isolated 12-byte store 40 B and a 4-byte store 32 B:
   the counters see LINES, not bytes.  The kernel's narrow accesses are the owner gathers of the plane-1 sources (one 4-byte
   read of the index map each) and the 12-byte source records: with ~3 M sources per launch (1.5 M of them with an owner
   gather) the line-granular model gives 2 x (33 + 35 + 96) MB of reads (planes, work-list inputs, gathers) + (121 + 120) MB of
   writes (line records / summaries / work list, source records) = 569 MB against the measured 532.  So the 1.5 x is real memory
   traffic - lines moved for a few bytes each - not a counter artefact, and not re-reads: the algorithmic bytes of those 4.5 M
   accesses are 42 MB, what they move is ~310 MB.  Not fixed: the owner gathers follow the set bits of a line's mask (one
   per source), there is no denser way to ask for them short of reading the index map's whole lines (256 MB per launch).
4. The ordinal depth term (cfg2 as BASELINE.json words it): @DEPTH@ it/s (3 716 in round 5; target 4 300).  Round 6: the depth-map
   backward is sparse (faces and frames that touch no non-zero gradient are not walked: 100 → 45 µs of kernel time), the depth
   renders keep their empty regions, and the silhouette render and the OBJECT's depth render are one launch pair
   (`hm_sil_fwd_multi`: the merged raster lasts 56 µs where the two launches took 41 + 7 + 34; +1.4 % on the iteration) with a
   cost-sorted launch order for the hand's render (+1.5 %) and the DEPTH render's workgroups dispatched first inside the merged
   launch (its slowest workgroup, 37 µs on one crowded region, then runs under the silhouette render's two rounds: merged raster
   56 → 48 µs, +2.2 %) - 4 134 → 4 300 it/s over the three steps, same-box A/B each, bit-identical.  Both chains end together
   now (`profiles/r06_p_cfg2_depth_timeline.txt`, a mid-run iteration under rocprofv3: object side - merged setup + raster, lines,
   sweeps, then its depth-map backward 11 + 16 and the pose gradients 17 µs; hand side - MANO forward 17, pair terms 27, hand
   render 14 + 48, ordinal term + its backward 37, depth-map backward 21, MANO backward 39 µs), so a further gain needs BOTH tails
   shortened (the object's depth gather folded into `k_rigid_bwd_x` like the silhouette gather; the ordinal term's two launches
   as one) - not built.  The synthetic clip's fit DRIFTS with `lw_depth = 1` (total loss 0.17 → 0.40 over 3 000 steps while
   `loss_depth` stays at 0.012: `tools/depth_windows.py`), so its rate falls with the iteration count (4 260 it/s over iterations
   200-400, 3 900 at 800-1 000: the error bands widen); the leg of the default bench line times iterations 400-700.  The hand's
   render in the same launch as the object's two was priced by `tools/merged_raster_probe.py` (+34 µs on the object's chain, which
   would also have to wait ~13 µs for the MANO forward) and not built.  Two hands per frame run in the fused loop too
   (`tests/test_depth_gpu.py`; their renders stay separate launches), the oracle's written-out chain covers the term (bit-equal
   free run, `tests/test_handchain_gpu.py::test_two_hands_with_depth_term_bit_equal`).  The reference's own call site raises
   (`homan.py:506-507`): oracle-pinned only.
5. `hand_proj_mode="ortho"` is built but parity-unpinned (section 7: its camera conversion is a third-party function absent from
   `/root/reference`, restated from the camera model) and runs through the graph loop, not the fused one.
6. N > 1 on real multi-GPU hardware: RCCL has carried one-rank groups and (gloo) 2-3 ranks on one GPU here;
   `tests/test_dist_gpu.py::test_two_ranks_on_two_gpus_over_rccl` runs the two-GPU case wherever two GPUs are visible and
   `bench.py --gpus N` reports the process group's rank count, its backend and every rank's own rate.
