"""FusedStepper: the whole optimisation iteration (reference homan/jointopt.py:158-192 over homan/homan.py:421-508) as a fixed
sequence of C-ABI kernel launches on two or three HIP streams, captured in hipGraphs - the benchmark path.  One clip, a batch
of equal-shaped clips (homan_amd.clipbatch), or - through homan_amd.shard - a shard of clips of any shapes."""
import ctypes
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import lib as _lib
from .homan import HOMan
from .loopcommon import HmAdam, parameter_groups


def _morton_order(verts):
    """permutation of the (V,3) vertices along a 3-D Morton curve (10 bits per axis): consecutive vertices are neighbours"""
    v = verts.detach().float().cpu().numpy()
    lo, hi = v.min(0), v.max(0)
    q = np.clip(((v - lo) / np.maximum(hi - lo, 1e-12) * 1023.0).astype(np.int64), 0, 1023)
    code = np.zeros(len(v), np.int64)
    for bit in range(10):
        for ax in range(3):
            code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
    return torch.from_numpy(np.argsort(code, kind="stable").astype(np.int32))


_LOOP_STREAMS = {}


def _loop_streams(device):
    """(capture, side, aux, warm-up) HIP streams of the fused loop, one set per device, pairwise distinct.
    torch hands streams out of a pool of 32 per device, round-robin: a process that builds many steppers eventually draws
    a side stream that IS the capture stream, and a capture in which two 'streams' wait on each other both ways crashes
    the HIP graph runtime at replay.  Steppers replay in stream order anyway, so they share one verified set."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _LOOP_STREAMS:
        got, seen = [], {torch.cuda.current_stream(key).cuda_stream, torch.cuda.default_stream(key).cuda_stream}
        for _ in range(256):
            st = torch.cuda.Stream(device=key)
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                got.append(st)
            if len(got) == 4:
                break
        assert len(got) == 4, "could not obtain four distinct HIP streams"
        _LOOP_STREAMS[key] = tuple(got)
    return _LOOP_STREAMS[key]


class FusedStepper:
    """The whole optimisation iteration as a fixed sequence of C-ABI kernel launches (no autograd tape, no torch
    arithmetic kernels), captured in a hipGraph:

        forward   rigid(obj) . mano . rigid(hand) . priors . smooth x2 . [collision] . [nn] . [contact] . v2d .
                  silhouettes(project, setup, raster, reduce) . inter . log(total + loss_evolution row)
        backward  sil(masks, sweeps, gather) . inter . weighted sums of the per-loss vertex gradients .
                  rigid_bwd(obj) . rigid_bwd(hand) . mano_bwd . prior terms . Adam

    Same kernels, same detach structure and same weighting as HOMan.forward + autograd (reference homan/homan.py:421-508,
    jointopt.py:178-192); tests/test_model_gpu.py checks the two paths produce the same gradients.  Supports the
    configurations of BASELINE.json (one right hand, optimize_mano=True, optimize_mano_beta=True, persp); anything
    else should use mode="graph".

    `model` is one HOMan, or a list of HOMan of identical shapes = a CLIP BATCH (homan_amd.clipbatch): the C clips are
    then optimised together, every kernel launched once over the C * B frames through the `*_clips` entry points
    (per-clip normalisers, sums, scales, Adam state and log rows; BASELINE cfg4).  Per clip the arithmetic - including
    the order of every floating-point sum - is the single-clip one, so a batched step equals C single steps bit for bit.
    With `shared_scale` (BASELINE cfg5) the object scale is ONE scalar for all clips of all ranks: each clip keeps a
    replica, the replicas' gradients are summed over the local clips and all-reduced over `group` once per iteration
    (one 4-byte RCCL all-reduce on the compute stream between the two captured halves of the iteration), and every
    replica takes the identical Adam step."""

    SLOTS = ["loss_pca", "loss_scale_obj", "loss_scale_hand", "loss_smooth_obj", "loss_smooth_hand", "loss_collision",
             "loss_contact", "loss_v2d_hand", "v2d_hand", "loss_sil_obj", "iou_object", "loss_inter",
             "handobj_maxdist", "loss_depth"]

    def __init__(self, model, loss_weights, lr, max_steps, capture=True, shared_scale=False, group=None, collectives=True):
        from . import constants, ops
        from .clipbatch import ClipBatch, ClipReduceWorkspace
        for one in (model.models if isinstance(model, ClipBatch) else model if isinstance(model, (list, tuple)) else [model]):
            if len(one.hand_sides) not in (1, 2):
                raise NotImplementedError("one or two hands per frame")
        m = self.model = model if isinstance(model, ClipBatch) else ClipBatch(model if isinstance(model, (list, tuple))
                                                                             else [model])
        if len({tuple(one.hand_sides) for one in m.models}) != 1:
            raise NotImplementedError("the clips of a batch share the hand side (one MANO model per launch)")
        if m.int_scales_hand.requires_grad or m.hand_proj_mode != "persp":
            raise NotImplementedError("FusedStepper covers optimize_mano_beta=True (the hand scale a buffer) and persp")
        self.h = h = len(m.models[0].hand_sides)
        kinds = {one.losses.inter_type for one in m.models}
        # inter_type "min" (reference losses.py:219-221, a HOMan option its loop cannot select): one hand, one clip
        self.inter_min = kinds == {"min"}
        if len(kinds) != 1 or (self.inter_min and (h > 1 or m.C > 1)):
            raise NotImplementedError("inter_type='min' in the fused loop: one hand, one clip (else mode='graph' or 'eager')")
        if h > 1 and (m.C > 1 or shared_scale):
            raise NotImplementedError("two hands per frame: the fused loop takes one clip at a time (a batch of two-hand "
                                      "clips: one stepper per clip, or mode='graph')")
        lw = self.lw = {k: float(v) for k, v in loss_weights.items()}
        if lw.get("lw_depth", 0) > 0:
            if not getattr(m, "ordinal_depth", False):
                # reference homan.py:506-507 calls lossutils.compute_ordinal_depth_loss() without its arguments
                raise TypeError("compute_ordinal_depth_loss() missing 3 required positional arguments: "
                                "'masks', 'silhouettes', and 'depths'")
        self.shared_scale, self.group = bool(shared_scale), group
        # collectives=False: the caller (ShardStepper: several steppers of one rank) issues the broadcast / all-reduce of the
        # tied scale itself - every rank must issue the same number of collectives whatever its number of steppers
        self.collectives = bool(collectives)
        if self.shared_scale and not m.optimize_object_scale:
            raise ValueError("shared_scale needs models built with optimize_object_scale=True")
        self.L, self.c, self.ops = _lib.lib(), constants, ops
        dev = m.translations_object.device
        B, Vo, Vh = m.B, m.verts_object_og.shape[1], 778          # B = frames of the whole batch
        C, NS = m.C, len(self.SLOTS) + 1
        self.B, self.C, self.clip_len, self.NS = B, C, m.clip_len, NS
        # third stream for the silhouette reduction + log row: it pays on a clip batch (+1.5 %); at one clip the graph executor
        # spends two cross-queue hops (~10 us each) on it, and two streams are 5-6 % faster (same-box A/B, cfg2 and cfg3)
        self.use_aux = (os.environ.get("HOMAN_AUX") or ("1" if C > 1 else "0")) != "0"
        if lw.get("lw_depth", 0) > 0:
            self.use_aux = False         # (with the depth launches on the side stream the three-stream graph dies at replay in
                                         #  the HIP runtime, like the other patterns listed at _loop_streams: two streams)
        # two streams, no shared scale: the log row of a step is written by the Adam launch itself (one launch less)
        self.log_in_adam = (not self.use_aux and not self.shared_scale and
                            os.environ.get("HOMAN_LOG_IN_ADAM", "1") != "0")
        self.fork_after_setup = os.environ.get("HOMAN_FORK_AFTER_SETUP", "1") != "0"
        # the depth renders write into buffers this stepper owns and keeps passing: a region that is empty again leaves the empty
        # pattern it wrote last time alone (hm_sil_fwd's persistent_outputs, as the silhouette render does) - at the full-image
        # camera nine regions in ten are background
        self.depth_persistent = int(os.environ.get("HOMAN_DEPTH_PERSISTENT", "1") != "0")
        # the side stream forms the camera-space object vertices ITSELF (hm_rigid_fwd_clips into a buffer of its own: the face
        # setup's arithmetic, the same floats) instead of waiting for the face setup's copy: the silhouette chain then has no
        # successor on another queue between the iteration's fork and its join - the rasteriser follows the face setup without
        # the few microseconds a node with a cross-queue successor costs its own queue (EXPERIMENTS r6)
        # (not on the step-2 sets: there the hand-side chain is the longer one, and one more launch on it costs what the
        #  silhouette chain gains - cfg3 5 457 -> 5 221 it/s with it)
        self.side_own_vo = ((os.environ.get("HOMAN_SIDE_OWN_VO") or "1") != "0" and self.h == 1 and C == 1 and
                            not lw.get("lw_depth", 0) > 0 and not lw.get("lw_collision", 0) > 0 and not lw.get("lw_contact", 0) > 0)
        # with the ordinal depth term the object's two renders of an iteration - ROI silhouette, full-image depth - are ONE launch
        # pair (hm_sil_fwd_multi; see _issue_silhouette_chain)
        self.merge_renders = (os.environ.get("HOMAN_MERGE_RENDERS", "1") != "0" and self.h == 1 and lw.get("lw_depth", 0) > 0 and
                              lw.get("lw_sil_obj", 0) > 0)
        # the silhouette loss / IoU values (log only) come out of the backward's first launch: one launch less on the chain
        # (two streams only: with the third stream the reduction and the log row stay there, behind the raster's event)
        self.sil_reduce_in_bwd = not self.use_aux and os.environ.get("HOMAN_SIL_REDUCE_IN_BWD", "1") != "0"
        self.Vo, self.Vh, self.P = Vo, Vh, m.mano_pca_pose.shape[1]
        f = lambda *shape: torch.zeros(*shape, device=dev)
        N = self.N = B * h                                        # hand rows (hands interleaved frame-major, homan.py:62-63)
        self.vo, self.vm, self.vh = f(B, Vo, 3), f(N, Vh, 3), f(N, Vh, 3)
        # the side stream's view of the object's vertices (see side_own_vo)
        self.vo_b = torch.zeros_like(self.vo) if self.side_own_vo else self.vo
        self.vals = f(C, NS)                                      # row c: the loss / metric slots of clip c + its total
        on = lambda k: lw.get(k, 0.0) > 0
        self.on = dict(pca=on("lw_pca"), so=on("lw_scale_obj"), sh=on("lw_scale_hand"),
                       smooth=on("lw_smooth_hand") or on("lw_smooth_obj"), col=on("lw_collision"),
                       con=on("lw_contact"), v2d=on("lw_v2d_hand"), sil=on("lw_sil_obj"), inter=on("lw_inter"),
                       depth=on("lw_depth"))
        w = {"loss_pca": lw["lw_pca"] if self.on["pca"] else 0, "loss_scale_obj": lw["lw_scale_obj"] if self.on["so"] else 0,
             "loss_scale_hand": lw["lw_scale_hand"] if self.on["sh"] else 0,
             "loss_smooth_obj": lw["lw_smooth_obj"] if self.on["smooth"] else 0,
             "loss_smooth_hand": lw["lw_smooth_hand"] if self.on["smooth"] else 0,
             "loss_collision": lw["lw_collision"] if self.on["col"] else 0,
             "loss_contact": lw["lw_contact"] if self.on["con"] else 0,
             "loss_v2d_hand": lw["lw_v2d_hand"] if self.on["v2d"] else 0,
             "loss_sil_obj": lw["lw_sil_obj"] if self.on["sil"] else 0,
             "loss_inter": lw["lw_inter"] if self.on["inter"] else 0,
             "loss_depth": lw["lw_depth"] if self.on["depth"] else 0}
        self.w = w
        self.weights = torch.tensor([w.get(k, 0.0) for k in self.SLOTS], device=dev)
        self.keys = [k for k in self.SLOTS if self._reported(k)]
        # the object's smoothness VALUE (its gradient is formed inside the rigid backward) rides the pair-terms launch of the side
        # stream as a block range.  Until round 6 the step-2 sets launched it on the silhouette chain instead, behind the sweeps
        # (round 3: cfg3 +5 % while the hand side was by far the longer chain); since the hand side's launches were fused the
        # silhouette chain is the longer one there too and an 8 us launch on its tail costs what it lasts: cfg3 5 405 -> 5 510
        # it/s over iterations 20-420, 5 246 -> 5 304 in the steady state (same box, alternated).  HOMAN_SMOOTH_OBJ_MAIN=1: the old
        # placement (one clip only: -4 % on a clip batch).
        self.smooth_obj_on_main = (os.environ.get("HOMAN_SMOOTH_OBJ_MAIN", "0") != "0" and
                                   (self.on["col"] or self.on["con"]) and m.C == 1)
        # unit gradients / scratch
        self.U_pca, self.U_so, self.U_sh = f(N, self.P), f(C), f(C)
        self.U_smo, self.U_smh, self.U_v2d = f(B, Vo, 3), f(N, Vh, 3), f(N, Vh, 3)
        self.U_colh, self.U_colo, self.U_conh, self.U_cono = f(N, Vh, 3), f(B, Vo, 3), f(N, Vh, 3), f(B, Vo, 3)
        self.G_sil, self.G_int_h, self.G_int_o = f(B, Vo, 3), f(N, Vh, 3), f(B, Vo, 3)
        self.G_o, self.G_h, self.G_mesh = f(B, Vo, 3), f(N, Vh, 3), f(N, Vh, 3)
        self.g_pca_mano, self.g_so_part = f(N, self.P), f(B)
        self.rec = f(N, 8)
        if self.inter_min:
            self.G_min_h, self.tmp_inter, self.rows = f(N, Vh, 3), f(2), torch.arange(B, device=dev)
        if h > 1:
            # two hands: the pair-wise terms see one hand at a time as a dense (B,778,3) array (hand i = rows i::h)
            self.vh_d = [f(B, Vh, 3) for _ in range(h)]
            self.nn_idx_d = [torch.zeros(B, Vh, dtype=torch.int32, device=dev) for _ in range(h)]
            self.nn_d2_d = [f(B, Vh) for _ in range(h)]
            self.U_conh_d, self.U_cono_d = [f(B, Vh, 3) for _ in range(h)], [f(B, Vo, 3) for _ in range(h)]
            self.U_colh2, self.U_col_d = f(N, Vh, 3), [f(B, Vh, 3) for _ in range(4)]      # (h0|h1), (h0|obj), (h1|obj): hand sides
            self.U_colo_d = f(B, Vo, 3)
            self.rec_d, self.G_int_o_d = [f(B, 8) for _ in range(h)], [f(B, Vo, 3) for _ in range(h)]
            self.tmp_h, self.tmp_col = f(h, 4), f(3)
            self.rws_h = [ClipReduceWorkspace(dev, 1) for _ in range(h)]
            self.hand_ctx = [m.mano_model.ctx_mean if sd == "right" else m.mano_model._left_ctx(False)
                             for sd in m.models[0].hand_sides]
        self.nn_idx = torch.zeros(B, Vh, dtype=torch.int32, device=dev)
        # the metric-only search's seed pairs (per frame the vertex pair that held the minimum at the last iteration: its distance
        # now bounds this iteration's minimum before anything is scanned; hm_nn_fwd_rigid_clips).  Scheduling data only.
        self.nn_seed = (torch.zeros((2 + (Vh + 127) // 128) * B, dtype=torch.int32, device=dev)
                        if os.environ.get("HOMAN_NN_SEED", "1") != "0" else None)
        self.nn_d2 = f(B, Vh)
        self.obj_order = _morton_order(m.verts_object_og[0]).to(dev)      # spatial sort of the rigid mesh (metric-only search)
        # ... and of the hand: its template's vertices in the same kind of order, so that the 128 hand vertices of a search
        # workgroup are a patch of the hand, not a sample of all of it (articulation moves the patches, it does not mix them)
        side = m.models[0].hand_sides[0]
        self.mctx = m.mano_model.ctx_mean if side == "right" else m.mano_model._left_ctx(False)      # (see ManoModel)
        self.hand_order = _morton_order(self.mctx.tensors[0]).to(dev)
        # bounding spheres, in MESH space, of the groups of 64 vertices in that order, per frame (a clip's frames share one mesh,
        # the clips of a batch need not): centre = mean, radius = farthest vertex.  The search carries them into camera space
        # with the frame's rigid transform instead of reducing the transformed vertices of every group in every workgroup.
        self.Vo, self.B = Vo, B
        with torch.no_grad():
            # (repeats of a real vertex pad the last group: they change nothing but the mean, and are masked out of it)
            self.obj_spheres = self._group_spheres()                                         # (B, ng, 4)
        self.nn_spheres = os.environ.get("HOMAN_NN_SPHERES", "1") != "0"
        self.pooled = f(B, m.sil_ctx.S, m.sil_ctx.S)
        # silhouettes at a size off the kernels' 32-pixel tile grid (the reference's REND_SIZE is 256): rendered on the next
        # multiple with the first two rows of K rescaled and the masks padded with keep = 0 (ops.SilhouetteContext); all three
        # are constants of the fit, built once; eps of the pseudo-gradient in the padded grid's NDC units
        sx = m.sil_ctx
        self.sil_K = sx.K_eff(m.camintr_rois_object).contiguous()
        self.sil_keep, self.sil_ref = sx.pad(m.keep_mask_object), sx.pad(m.ref_mask_object)
        self.sil_eps = sx.eps()
        self.up_sil, self.up_inter = torch.tensor([w["loss_sil_obj"]], device=dev), torch.tensor([w["loss_inter"]], device=dev)
        if self.on["depth"] and h > 1:
            # two hands per frame: the three layers [object, hand 0, hand 1] of reference homan.py:384-419, every unordered pair
            # through the two-layer kernels, one normaliser for the scene (lossutils.py:133-169; ops.ordinal_depth_loss_layers is
            # the autograd form of what _forward_backward_hands issues)
            ctx_o, ctx_hs, m_o, m_hs = m.models[0]._depth_contexts_hands()
            if ctx_o.padded:
                raise NotImplementedError("the fused loop renders the depth images at image_size % 32 == 0; other sizes: "
                                          "mode='graph' or 'eager'")
            self.dlayers = [(ctx_o, Vo, m_o)] + [(ctx_hs[i], Vh, m_hs[i]) for i in range(h)]
            self.dpairs = [(a, b) for a in range(h + 1) for b in range(a + 1, h + 1)]
            Sd = ctx_o.S
            self.dl_sil, self.dl_dep = [f(B, Sd, Sd) for _ in self.dlayers], [f(B, Sd, Sd) for _ in self.dlayers]
            self.dl_g = [f(B, Sd, Sd) for _ in self.dlayers]
            self.dp_part, self.dp_rec = [f(B * 8) for _ in self.dpairs], [f(8) for _ in self.dpairs]
            self.dp_out, self.dp_up = [f(1) for _ in self.dpairs], [f(1) for _ in self.dpairs]
            self.dp_g = [(f(B, Sd, Sd), f(B, Sd, Sd)) for _ in self.dpairs]
            self.rws_dp = [ClipReduceWorkspace(dev, 1) for _ in self.dpairs]
            self.G_dep_o, self.G_dep_h_d, self.G_dep_h = f(B, Vo, 3), [f(B, Vh, 3) for _ in range(h)], f(N, Vh, 3)
        elif self.on["depth"]:
            # ordinal depth term (reference homan.py:384-419, opt-in): object and hand rendered with depth at the full-image
            # camera, the pair-wise ordinal loss, and its gradient back through both depth images to the camera-space vertices
            if C == 1:
                self.dctx = m.models[0].depth_contexts()
            else:
                # a clip batch: the two depth renders run over all frames at once, the ordinal term (it normalises over ONE
                # clip: pairs, mask counts) per clip on its slice of the images
                m0_, size = m.models[0], int(m.image_size)
                for one in m.models:
                    if tuple(one.masks_object.shape[1:]) != (size, size) or tuple(one.masks_human.shape[1:]) != (size, size):
                        raise NotImplementedError("ordinal depth: instance masks must be (B, image_size, image_size)")
                self.dctx = (ops.SilhouetteContext(m0_.faces_object[:1].expand(B, -1, -1), Vo, B, size, dev),
                             ops.SilhouetteContext(m0_.faces_hand[:1].expand(B, -1, -1), 778, B, size, dev),
                             torch.cat([(one.masks_object != 0).to(torch.uint8) for one in m.models]).contiguous(),
                             torch.cat([(one.masks_human != 0).to(torch.uint8) for one in m.models]).contiguous())
            ctx_o, ctx_h = self.dctx[0], self.dctx[1]
            if ctx_o.padded:
                raise NotImplementedError("the fused loop renders the depth images at image_size % 32 == 0; other sizes: "
                                          "mode='graph' or 'eager'")
            Sd = ctx_o.S
            self.d_sil_o, self.d_dep_o, self.d_sil_h, self.d_dep_h = f(B, Sd, Sd), f(B, Sd, Sd), f(B, Sd, Sd), f(B, Sd, Sd)
            self.d_go, self.d_gh, self.d_part, self.d_rec = f(B, Sd, Sd), f(B, Sd, Sd), f(B * 8), f(C, 8)
            # non-zero structure of the two gradient images, one byte per (frame, pixel row, 64-pixel segment): the depth-map
            # backward skips faces and frames that touch no flagged segment (the term is zero wherever render and annotation
            # agree on the order); ones = "walk everything" until the first backward has written them
            self.d_flags = (torch.ones(2, B * Sd * (Sd // 64), dtype=torch.uint8, device=dev)
                            if Sd % 64 == 0 and os.environ.get("HOMAN_DEPTH_SPARSE", "1") != "0" else None)
            self.rws_depth = ClipReduceWorkspace(dev, C)
            self.G_dep_o, self.G_dep_h = f(B, Vo, 3), f(B, Vh, 3)
            self.up_depth = torch.tensor([w["loss_depth"]], device=dev)
        # static gradient buffers for exactly the parameters that receive gradients in this configuration
        for p in m.parameters():
            p.grad = None
        gp = [m.translations_object, m.rotations_object, m.translations_hand, m.rotations_hand]
        if m.optimize_mano:          # (optimize_mano=False, the reference function's own default: the hand mesh is the
            gp += [m.mano_pca_pose, m.mano_rot, m.mano_trans, m.mano_betas]      # constant `verts_hand_og`, homan.py:357-358)
        if m.optimize_object_scale:
            gp.append(m.int_scales_object)
        for p in gp:
            p.grad = torch.zeros_like(p)
        self._build_optimizers(lr)
        self.log_buf = torch.zeros(max_steps, C, NS, device=dev)
        self.max_steps = max_steps
        self.rigid_ws_h, self.rigid_ws_o = (torch.zeros(self.L.hm_rigid_workspace_bytes(n), dtype=torch.uint8, device=dev)
                                            for n in (N, B))
        self.mano_state = torch.empty(self.L.hm_mano_state_bytes(N), dtype=torch.uint8, device=dev)
        self.graph = self.graph_b = self.graph_k = None
        # iterations per replay of the second graph (run()): one clip, no collective between the halves of an iteration.  The
        # turnaround between two replays is ~5 us of a 160 us iteration (same-box A/B: +2-3 % at 4, no more at 8 / 16)
        self.graph_iters = int(os.environ.get("HOMAN_GRAPH_ITERS") or ("4" if C == 1 else "1"))
        self.cap_stream, self.side, self.aux, side = _loop_streams(dev)
        self.ev_vo, self.ev_pair, self.ev_sil, self.ev_fwd, self.ev_smo, self.ev_ras = (torch.cuda.Event() for _ in range(6))
        self.ev_hand, self.ev_col, self.ev_dep, self.ev_dgrad = (torch.cuda.Event() for _ in range(4))
        # (option, off: one clip, step-2 sets - the collision chain on the side stream, the rest of the hand side on the third
        #  stream, see forward_backward; measured +-0.4 % on cfg3: both chains already share a work-bound GPU)
        self.col_on_aux = (os.environ.get("HOMAN_COL_AUX") or "0") != "0" and C == 1 and self.h == 1
        self.reduce_ws_b = ClipReduceWorkspace(dev, C)
        # (the terms of the fused pair-terms launch run side by side: a reduce workspace each)
        self.reduce_ws_c, self.reduce_ws_d, self.reduce_ws_e = (ClipReduceWorkspace(dev, C) for _ in range(3))
        # (one clip: the hand-side chain is the iteration's critical path, -6 %; a batch hides that chain under the silhouette
        #  chain and the fused launch only adds contention there, +1.6 %)
        # a clip batch: the pair-wise terms wait for the END of the rasteriser.  They used to start there anyway, behind a MANO
        # forward as long as the raster; since that launch reads the blend matrix once per four frames it is over early, and the
        # search / smoothness / interaction launches next to the raster cost it more (362 -> 433 us) than they gain next to the
        # line expansion (230 -> 162 us)
        self.pairs_after_raster = (self.use_aux and not self.sil_reduce_in_bwd and
                                   (os.environ.get("HOMAN_PAIRS_AFTER_RASTER") or "1") != "0")
        # a clip batch: the pair-wise terms of the side stream wait for the END of the line expansion (the backward in two
        # calls) - that kernel is latency-bound and takes 200 us instead of 150 next to neighbours that hold its wave slots -
        # and the sweeps run 1024 persistent workgroups instead of 1280 so that the hand's gradient launches find registers
        # next to them (same-box A/B, 8 clips: step-1 8 650 -> 8 865 it/s, step-2 7 142 -> 7 332; either change alone loses)
        self.pairs_after_lines = (os.environ.get("HOMAN_PAIRS_AFTER_LINES") or "1") != "0" and C > 1 and self.on["sil"]
        self.ev_lines = torch.cuda.Event()
        # the hand's rigid backward inside the MANO backward's launch (hm_mano_bwd_rigid_clips): one launch less on the hand-side
        # chain.  One clip: cfg2 +1.3 %, cfg3 +1.4 %; a clip batch hides that chain under the silhouette chain and loses 1-1.6 %
        # (round 6: a clip batch too, +0.5 % - the separate hand launch was a 1024-thread workgroup per frame that found no room
        #  next to the persistent sweeps: 106 us on average, up to 365 us in the 8-clip profile of round 5)
        #  (a step-2 batch keeps the separate launch: 7 715 against 7 647 it/s)
        self.mano_bwd_rigid = (os.environ.get("HOMAN_MANO_BWD_RIGID") or
                               ("1" if C == 1 or not (self.on["col"] or self.on["con"]) else "0")) != "0"
        self.nn_early = (os.environ.get("HOMAN_NN_EARLY") or "0") != "0"
        self.pair_fused = (os.environ.get("HOMAN_PAIR_FUSED") or ("1" if C == 1 else "0")) != "0"
        self.hand_terms_fused = os.environ.get("HOMAN_HT_FUSED", "1") != "0"
        self.nn_full_fused = os.environ.get("HOMAN_NN_FULL_FUSED", "1") != "0"
        if self.shared_scale:
            self._sync_shared_scale_start()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.forward_backward()              # warm-up, no optimiser step
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.on["sil"]:
            m.sil_ctx.calibrate()                # cost-sorted launch orders from the current state (scheduling only)
        if self.on["depth"] and self.h == 1:
            # the HAND's depth render gets a cost-sorted launch order and its near-winding hint too (its active workgroups started
            # up to 32 us late behind background regions of the static order, profiles/r06_raster_trace_depth.txt; with the
            # object's two renders in one launch the hand side is the longer chain: cfg2 + depth 4 151 -> 4 215 it/s, same box);
            # the object's depth render keeps the static order (sorted: +-0 alone, -1 % with the hand's)
            for tag, ctx in zip("oh", self.dctx[:2]):
                if tag in os.environ.get("HOMAN_DEPTH_CALIBRATE", "h"):
                    ctx.calibrate()
        if capture:
            # scheduling hint baked into the captured launches: with the collision / contact terms the hand-side stream is
            # the longer chain and the persistent edge sweeps should leave it more of the GPU (same results either way;
            # same-box A/B on cfg3, round 2 before the launch fusions: 1280 -> 4030, 768 -> 4130, 512 -> 4194 it/s; after them,
            # with the raster ballast below: 512 -> 4408, 768 -> 4552, 1024 -> 4537, 1280 -> 4512)
            sb = int(os.environ.get("HOMAN_SWEEP_BLOCKS", "0")) or (768 if (self.on["col"] or self.on["con"]) and C == 1 else
                                                                    1024 if self.pairs_after_lines else 1280)
            pad = os.environ.get("HOMAN_RASTER_PAD")
            # same idea for the rasteriser: 8 KB of LDS ballast = 4 workgroups per CU instead of 6 leaves registers for the
            # hand-side kernels (cfg3 one clip: 4347 -> 4500 it/s; cfg2, where that chain is short: 5820 -> 5500, so not there)
            pad = int(pad) if pad is not None else (4096 if (self.on["col"] or self.on["con"]) and C == 1 else 0)
            # the raster's launch order follows the measured cost of its workgroups from iteration to iteration
            # (hm_tune_raster_reorder; same-box A/B: raster 57 -> 45 us inside the graph at one clip, 356 -> 298 us at eight; the
            # iteration: clip batches and the step-2 sets +0.3..1 %, one-clip step-1 fits +4 % in the steady state and over
            # iterations 5-25 - but only since the metric-only search got shorter: while the hand-side chain was as long as the
            # silhouette chain it had been running in the raster's tail and a shorter raster pushed it under the sweeps, -4 %)
            ro = int(os.environ.get("HOMAN_RASTER_REORDER", "1"))
            # and for the metric-only search of a clip batch: its 1680 small, latency-bound workgroups otherwise take every wave
            # slot of the CUs next to the line expansion (lines 250 -> 226 us, iteration -4.4 % at 3 search workgroups per CU;
            # 2 per CU make the search itself the tail)
            nn_pad = os.environ.get("HOMAN_NN_PAD")
            # (34 KB: with the sweep's 30.5 KB workgroups a CU then holds 4 sweeps + 1 search, 2 + 2 or 1 + 3 - the mixes the
            #  40 KB ballast gave next to the 28.9 KB workgroups of the float sweeps; at 40 KB the search found room only
            #  next to THREE sweep workgroups and took 297 instead of 137 us, same-box profile)
            nn_pad = int(nn_pad) if nn_pad is not None else (34816 if C > 1 and not self.on["con"] else 0)
            fam_pads = [int(x) for x in os.environ.get("HOMAN_FAM_PADS", "0,0,0,0,0").split(",")]
            # the hints are per-thread values read when a launch is issued (= captured): set, capture, restore - whatever
            # happens in between (a capture that raises must not leave them changed for the next stepper)
            tune = _lib.lib()
            prev = tune.hm_tune_sweep_blocks(sb)
            prev_pad = tune.hm_tune_raster_lds_pad(pad)
            prev_reorder = tune.hm_tune_raster_reorder(ro)
            prev_nn_pad = tune.hm_tune_nn_lds_pad(nn_pad)
            prev_fam = [tune.hm_tune_lds_pad(i, v) for i, v in enumerate(fam_pads)]
            prev_rc = tune.hm_tune_rigid_chunked(int(os.environ.get("HOMAN_RIGID_CHUNKED", "1")))
            try:
                self.graph = _lib.new_graph()
                with torch.cuda.graph(self.graph, stream=self.cap_stream):
                    self.forward_backward(log=not self.log_in_adam)
                    if not self.shared_scale:
                        self._capture_step()
                if self.shared_scale:        # the all-reduce of the scale gradient runs between two captured halves
                    self.graph_b = _lib.new_graph()
                    with torch.cuda.graph(self.graph_b, stream=self.cap_stream):
                        self._spread_shared_scale_grad()
                        self.opt.step(zero_grad=False)
                elif self.graph_iters > 1:
                    # K iterations in ONE graph: the boundary between two replays (the executor's own start-up, a few
                    # microseconds of an iteration that lasts 160) is paid once per K iterations; run() replays this graph
                    # for every full K and the one-iteration graph for the rest - the same launches either way
                    self.graph_k = _lib.new_graph()
                    with torch.cuda.graph(self.graph_k, stream=self.cap_stream):
                        self._capture_iterations(self.graph_iters)
            finally:
                tune.hm_tune_sweep_blocks(prev)
                tune.hm_tune_raster_lds_pad(prev_pad)
                tune.hm_tune_raster_reorder(prev_reorder)
                tune.hm_tune_nn_lds_pad(prev_nn_pad)
                for i, v in enumerate(prev_fam):
                    tune.hm_tune_lds_pad(i, v)
                tune.hm_tune_rigid_chunked(prev_rc)

    # ---- other clips into the resident stepper
    def reload(self, clip_inputs):
        """`clip_inputs`: one dict of HOMan data arguments per clip of the batch (what `collate_inputs` returns, + camintr),
        clips of exactly the shapes and topology this stepper was built for.  Everything that depends on the clip is copied in
        place - Parameters, buffers, per-clip normalisers, the search's bounding spheres - and everything that depends on the
        FIT is reset - Adam moments, step counter, loss slots -; buffers, workspaces, streams and the captured hipGraph are
        reused as they are.  The next `run(steps)` is the fit of the new clips, bit-identical to the fit a freshly built
        stepper would make (tests/test_clip_fitter_gpu.py)."""
        m = self.model
        if len(clip_inputs) != m.C:
            raise ValueError(f"reload: {len(clip_inputs)} clips for a stepper of {m.C}")
        from .clipbatch import _PER_CLIP, _PER_FRAME
        with torch.no_grad():
            for c, (one, kw) in enumerate(zip(m.models, clip_inputs)):
                one.load_clip(**kw)
                if m.C > 1:     # the batch's own concatenated copies of the BUFFERS (Parameters are views of its storage)
                    for k in _PER_FRAME + _PER_CLIP:
                        if hasattr(m, k) and not isinstance(getattr(m, k), torch.nn.Parameter):
                            m.clip_slice(getattr(m, k), c).copy_(getattr(one, k))
                    m.keep_sum[c:c + 1].copy_(one.losses.keep_sum)
            if m.C > 1:
                m.sil_ctx.invalidate_outputs()
            sx = m.sil_ctx
            if sx.padded:       # (unpadded: these ARE the model's tensors)
                self.sil_K.copy_(sx.K_eff(m.camintr_rois_object))
                self.sil_keep.copy_(sx.pad(m.keep_mask_object))
                self.sil_ref.copy_(sx.pad(m.ref_mask_object))
            self.obj_spheres.copy_(self._group_spheres())
            if self.on["depth"] and self.h == 1 and m.C == 1:     # (two hands: the layers' masks are the model's own tensors, copied in place)
                self.dctx = m.models[0].depth_contexts()
            elif self.on["depth"] and self.h == 1:
                # a clip batch: the instance masks of the depth term are the stepper's own concatenation of the clips' masks
                CL = m.clip_len
                for c, one in enumerate(m.models):
                    self.dctx[2][c * CL:(c + 1) * CL].copy_((one.masks_object != 0).to(torch.uint8))
                    self.dctx[3][c * CL:(c + 1) * CL].copy_((one.masks_human != 0).to(torch.uint8))
            for st_m, st_v in self.opt.state:
                st_m.zero_()
                st_v.zero_()
            self.opt.step_t.zero_()
            self.vals.zero_()
            for p in m.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            if self.shared_scale:
                self._sync_shared_scale_start()

    def _group_spheres(self):
        """bounding spheres, in MESH space, of the groups of 64 object vertices in `obj_order`, per frame: (B, groups, 4)"""
        m, Vo, B = self.model, self.Vo, self.B
        ng = (Vo + 63) // 64
        vs = m.verts_object_og.detach()[:, self.obj_order.long()]
        pad = ng * 64 - Vo
        valid = torch.ones(Vo + pad, dtype=torch.bool, device=vs.device)
        if pad:
            vs = torch.cat([vs, vs[:, -1:].expand(-1, pad, -1)], 1)
            valid[Vo:] = False
        grp = vs.reshape(B, ng, 64, 3)
        wgt = valid.reshape(1, ng, 64, 1).float()
        ctr = (grp * wgt).sum(2) / wgt.sum(2)
        rad = ((grp - ctr[:, :, None]) ** 2).sum(-1).sqrt().amax(2)
        return torch.cat([ctr, rad[..., None]], -1).contiguous()

    # ---- shared object scale (BASELINE cfg5): C local replicas of one scalar, kept identical on every rank
    def _dist_on(self):
        import torch.distributed as dist
        return (self.collectives and dist.is_available() and dist.is_initialized() and
                dist.get_world_size(self.group) > 1)

    def _sync_shared_scale_start(self):
        import torch.distributed as dist
        s = self.model.int_scales_object
        with torch.no_grad():
            s0 = s.data[:1].clone()              # one element on the wire whatever the number of local clips
            if self._dist_on():
                from .dist import broadcast_shared_scalar, group_src
                broadcast_shared_scalar(s0, group_src(self.group), self.group)
            s.copy_(s0.expand_as(s))
        self.g_shared = torch.zeros(1, device=s.device)

    def _reduce_shared_scale_grad(self):
        """One fp32 per step over xGMI: sum over ranks of (sum over local clips of d loss / d scale), on the compute
        stream, no host synchronisation."""
        if self._dist_on():
            from .dist import sync_shared_scalar_grad
            sync_shared_scalar_grad(self.g_shared, self.group)

    def _spread_shared_scale_grad(self):
        # every replica receives the global sum (identical Adam steps keep the replicas bit-identical)
        g = self.model.int_scales_object.grad
        g.copy_(self.g_shared.expand_as(g))

    def _reported(self, k):
        o = self.on
        return {"loss_pca": o["pca"], "loss_scale_obj": o["so"], "loss_scale_hand": o["sh"], "loss_smooth_obj": o["smooth"],
                "loss_smooth_hand": o["smooth"], "loss_collision": o["col"], "loss_contact": o["con"],
                "loss_v2d_hand": o["v2d"], "v2d_hand": o["v2d"], "loss_sil_obj": o["sil"], "iou_object": o["sil"],
                "loss_inter": o["inter"], "handobj_maxdist": o["inter"], "loss_depth": o["depth"]}[k]

    def _slot(self, name):
        """device address of slot `name` of clip 0; clip c is `self.NS` floats further (the kernels' out_stride)"""
        i = self.SLOTS.index(name)
        return self.vals.data_ptr() + 4 * i

    def forward_backward(self, log=False):
        """Two concurrent branches (fork/join on HIP streams, captured as parallel branches of the hipGraph):
        A (calling stream): object transform, silhouettes forward/backward, object gradients;
        B (side stream):    MANO, hand transform, priors, 2-D / smoothness / collision / contact / interaction losses,
                            hand gradients, MANO backward.
        B waits for the object vertices before the pair-wise losses, A waits for B's object-side gradient terms, both
        join before the log row and the Adam step.  Every launch covers all the clips of the batch (clip_len frames
        each, per-clip scalars NS floats apart in `vals`).  The launches are ISSUED in the order of the builders below - the order
        is part of the design: the graph executor maps the captured branches onto its queues by it (EXPERIMENTS.md)."""
        if self.h > 1:
            return self._forward_backward_hands(log)
        it = self._iteration_namespace(log)
        main = it.main
        self.side.wait_stream(main)
        self._issue_silhouette_chain(it)
        with torch.cuda.stream(self.side):
            self._issue_hand_forward(it)
            # (issued in this order on purpose - with the depth term too: its renders BEFORE the pair-wise terms put the hand's
            #  raster next to the silhouette raster, cfg2 + depth 4 000 -> 3 790 it/s, EXPERIMENTS r6)
            self._issue_pair_terms(it)
            if self.on["depth"]:
                self._issue_depth_terms(it)
            self._issue_hand_backward(it)
        self._issue_object_backward(it)
        self._issue_join(it)

    # ---- hooks of the capture (tools/chain_only.py overrides them for its measurement-only launch graphs)
    def _build_optimizers(self, lr):
        self.opt = HmAdam(parameter_groups(self.model, lr))

    def _capture_step(self):
        """the optimiser step behind one captured iteration"""
        self.opt.step(zero_grad=False, log=self._adam_log())

    def _capture_iterations(self, K):
        """K iterations of the second graph"""
        for _ in range(K):
            self.forward_backward(log=not self.log_in_adam)
            self._capture_step()

    def _iteration_namespace(self, log=False):
        """what the per-chain issue builders share (forward_backward)"""
        from types import SimpleNamespace
        m, P = self.model, _lib.ptr
        main = torch.cuda.current_stream()
        return SimpleNamespace(m=m, L=self.L, P=P, ck=_lib.check, B=self.B, Vo=self.Vo, Vh=self.Vh, c=self.c, on=self.on, w=self.w,
                               CL=self.clip_len, NS=self.NS, C=self.C, main=main, side=self.side, sa=main.cuda_stream,
                               sb=self.side.cuda_stream, rws_a=P(m.reduce_ws.buf), rws_b=P(self.reduce_ws_b.buf), sctx=m.sil_ctx,
                               cctx=m.collision_ctx, pca=m.mano_pca_pose, rot=m.mano_rot, betas=m.mano_betas,
                               mtr=m.mano_trans if m.optimize_mano else None, npca=self.P * self.clip_len,      # PCA entries of one clip
                               use_aux=self.use_aux, log=log)

    def _aux_block(self, it):
        """the silhouette reduction and the log row on the third stream (clip batches)"""
        (m, L, P, ck, B, Vo, on, CL, NS, C, sctx, use_aux, log) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.on, it.CL, it.NS, it.C, it.sctx, it.use_aux, it.log)
        if not use_aux:
            return
        # the silhouette reduction and the log row run on a third stream, off both chains.  (Only this: HIP stream
        # capture crashes when two captured streams wait for each other's events in both directions, and the hipGraph
        # executor maps richer fork patterns onto its hardware queues in orders that serialise the branches -- both
        # measured.  WHERE this block is issued matters too: issued after the hand-side backward, the executor runs it
        # behind that chain, +20 us on the iteration.)
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(self.ev_fwd)
            if on["smooth"] and self.smooth_obj_on_main:
                self.aux.wait_event(self.ev_smo)         # (that loss value comes from the calling stream here)
            if on["sil"] and not self.sil_reduce_in_bwd:
                self.aux.wait_event(self.ev_ras)
                ck(L.hm_sil_reduce_clips(B, Vo, sctx.F, sctx.S, P(m.keep_sum), self._slot("loss_sil_obj"), None,
                                         P(sctx.workspace), CL, NS, self.aux.cuda_stream), "sil_reduce")
            if log:
                ck(L.hm_log_total_clips(P(self.vals), P(self.weights), len(self.SLOTS), P(self.opt.step_t),
                                        self.max_steps, P(self.log_buf), C, self.aux.cuda_stream), "log")

    def _issue_silhouette_chain(self, it):
        """A, first half: face setup (+ camera-space vertices), fork point, rasteriser, [object depth render], line expansion + sweeps"""
        (m, L, P, ck, B, Vo, on, CL, NS, main, sa, sctx, use_aux) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.on, it.CL, it.NS, it.main, it.sa, it.sctx, it.use_aux)
        # ---------------- A: silhouettes forward + backward (the critical chain: nothing else rides it; the object's rigid
        # transform is applied inside the face setup, the other losses get the vertices from the side stream)
        if on["sil"]:
            fwd_args = (P(m.verts_object_og), P(sctx.faces), 0, P(self.sil_K), B, Vo, sctx.F, sctx.S,
                        1.0, self.ops.NMR_NEAR, self.ops.NMR_FAR, P(self.sil_keep), P(self.sil_ref),
                        None, P(self.pooled), None, P(sctx.work_order), None, None, 0, P(m.rotations_object),
                        P(m.translations_object), P(m.int_scales_object), 1, 1, P(sctx.workspace), CL, NS, P(self.vo))
            if self.merge_renders:
                # with the ordinal depth term: the silhouette render and the OBJECT's depth render (the same mesh and pose at the
                # full-image camera) as ONE face-setup launch and ONE raster launch (hm_sil_fwd_multi: per render the arguments
                # of the two calls below, the depth render forming its vertices from the pose like the silhouette render does -
                # the same floats as self.vo); each render keeps its workspace, the backward passes are unchanged
                arr = _lib.sil_renders([
                    dict(verts=m.verts_object_og, faces=sctx.faces, K=self.sil_K, B=B, V=Vo, F=sctx.F, S=sctx.S, orig_size=1.0,
                         znear=self.ops.NMR_NEAR, zfar=self.ops.NMR_FAR, keep=self.sil_keep, ref=self.sil_ref, pooled=self.pooled,
                         work_order=sctx.work_order, rigid_rot6d=m.rotations_object, rigid_trans=m.translations_object,
                         rigid_scale=m.int_scales_object, rigid_abs=1, persistent_outputs=1, workspace=sctx.workspace, clip_len=CL,
                         cam_verts_out=self.vo),
                    dict(verts=m.verts_object_og, faces=self.dctx[0].faces, K=m.camintr, B=B, V=Vo, F=self.dctx[0].F,
                         S=self.dctx[0].S, orig_size=1.0, znear=self.ops.NMR_NEAR, zfar=self.ops.NMR_FAR, pooled=self.d_sil_o,
                         pooled_depth=self.d_dep_o, work_order=self.dctx[0].work_order, rigid_rot6d=m.rotations_object,
                         rigid_trans=m.translations_object, rigid_scale=m.int_scales_object, rigid_abs=1,
                         persistent_outputs=self.depth_persistent, workspace=self.dctx[0].workspace, clip_len=CL)])
                if os.environ.get("HOMAN_MERGE_ORDER", "1") == "1":
                    # the DEPTH render's workgroups first: it lasts as long as its slowest workgroup (37 us: one region with 240
                    # candidates, profiles/r06_raster_trace_depth.txt), which then runs under the silhouette render's two rounds
                    # instead of behind them (cfg2 + depth 4 208 -> 4 303 it/s, same box; results do not depend on the order)
                    tmp = _lib.SilRender()
                    ctypes.memmove(ctypes.byref(tmp), ctypes.byref(arr[0]), ctypes.sizeof(tmp))
                    ctypes.memmove(ctypes.byref(arr[0]), ctypes.byref(arr[1]), ctypes.sizeof(tmp))
                    ctypes.memmove(ctypes.byref(arr[1]), ctypes.byref(tmp), ctypes.sizeof(tmp))
                if self.fork_after_setup:
                    ck(L.hm_sil_fwd_multi(arr, 2, 1, sa), "sil_fwd_multi(setup)")
                    self.ev_sil.record(main)
                    ck(L.hm_sil_fwd_multi(arr, 2, 2, sa), "sil_fwd_multi(raster)")
                else:
                    ck(L.hm_sil_fwd_multi(arr, 2, 3, sa), "sil_fwd_multi")
                    self.ev_sil.record(main)
                self.ev_dep.record(main)
            elif self.side_own_vo:
                ck(L.hm_sil_fwd_clips(*fwd_args, sa), "sil_fwd")      # (no successor on the side stream: see side_own_vo)
            elif self.fork_after_setup:
                # the face setup (which also writes the camera-space vertices self.vo), the fork of the side stream, then the
                # rasteriser: the pair-wise losses do not wait for the raster and the raster has one successor on its chain
                ck(L.hm_sil_fwd_phase_clips(*fwd_args, 1, sa), "sil_fwd(setup)")
                self.ev_sil.record(main)
                ck(L.hm_sil_fwd_phase_clips(*fwd_args, 2, sa), "sil_fwd(raster)")
                if use_aux and not self.sil_reduce_in_bwd:
                    self.ev_ras.record(main)     # the tile partials of the fused loss, for the reduction on the third stream
            else:
                ck(L.hm_sil_fwd_clips(*fwd_args, sa), "sil_fwd")      # (also writes the camera-space vertices self.vo)
                self.ev_sil.record(main)         # self.vo for the side stream
                if use_aux and not self.sil_reduce_in_bwd:
                    self.ev_ras.record(main)
            if on["depth"] and not self.merge_renders:
                # the OBJECT's depth render of the ordinal depth term rides this chain, right behind the silhouette raster (its
                # vertices are the face setup's): the hand's render runs on the side stream meanwhile - two renders after each
                # other there made the hand side twice as long as this chain
                self._depth_render(self.vo, self.dctx[0], Vo, self.d_sil_o, self.d_dep_o, sa)
                self.ev_dep.record(main)
            bwd_args = (P(self.vo), P(self.sil_K), B, Vo, sctx.F, sctx.S, 1.0, self.sil_eps,
                        2 if self.lw["lw_sil_obj"] > 0 else 1,
                        P(self.up_sil), None, P(m.keep_sum), P(sctx.adj_off), P(sctx.adj_items),
                        P(sctx.face_order), None, None, P(sctx.workspace), CL,
                        self._slot("loss_sil_obj") if self.sil_reduce_in_bwd else None, NS)
            q2 = sctx.sum_log2q          # grid of the order-independent sums (the same for the sweeps and the rigid backward)
            # (no vertex gather; the loss / IoU values come out of its first launch)
            if self.pairs_after_lines:
                # a clip batch: the line expansion - latency-bound, and the kernel of this chain that suffers most from
                # neighbours holding its wave slots - runs ALONE; the pair-wise terms of the side stream wait for its end
                ck(L.hm_sil_bwd_phase_clips(*bwd_args, 1, q2, sa), "sil_bwd(lines)")
                self.ev_lines.record(main)
                ck(L.hm_sil_bwd_phase_clips(*bwd_args, 2, q2, sa), "sil_bwd(sweeps)")
            else:
                ck(L.hm_sil_bwd_clips(*bwd_args, q2, sa), "sil_bwd")

    def _issue_hand_forward(self, it):
        """B: object / hand vertices, the hand-only terms, then the waits that place the pair-wise terms next to the silhouette chain"""
        (m, L, P, ck, B, Vo, Vh, on, CL, NS, C, side, sb, rws_b, pca, rot, betas, mtr, npca) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.Vh, it.on, it.CL, it.NS, it.C, it.side, it.sb, it.rws_b, it.pca, it.rot, it.betas, it.mtr, it.npca)
        if not on["sil"]:    # (with the silhouette term the face setup of hm_sil_fwd has written self.vo already)
            ck(L.hm_rigid_fwd_clips(P(m.verts_object_og), P(m.rotations_object), P(m.translations_object),
                                    P(m.int_scales_object), 1, B, Vo, None, P(self.vo), CL, sb), "rigid_fwd(obj)")
            self.ev_vo.record(side)
        elif self.side_own_vo:
            ck(L.hm_rigid_fwd_clips(P(m.verts_object_og), P(m.rotations_object), P(m.translations_object),
                                    P(m.int_scales_object), 1, B, Vo, None, P(self.vo_b), CL, sb), "rigid_fwd(obj, side copy)")
        if m.optimize_mano:
            ck(L.hm_mano_fwd_clips(self.mctx.ptrs, P(pca), self.P, P(rot), P(betas), P(mtr), B, P(self.vm), None,
                                   P(m.rotations_hand), P(m.translations_hand), P(m.int_scales_hand), P(self.vh),
                                   P(self.mano_state), CL, sb),
               "mano_fwd + rigid(hand)")
        else:           # the hand mesh is the constant `verts_hand_og` (reference homan.py:357-358): rigid transform only
            ck(L.hm_rigid_fwd_clips(P(m.verts_hand_og), P(m.rotations_hand), P(m.translations_hand),
                                    P(m.int_scales_hand), 0, B, Vh, None, P(self.vh), CL, sb), "rigid_fwd(hand)")
        pri = on["pca"] or on["so"] or on["sh"]
        # pair terms that feed nothing to each other go in ONE launch (csrc/pairterms.hip): the interaction term, the
        # object's smoothness when it rides this stream, the metric-only search (no contact term) and the hand-only
        # reductions
        sm_here = on["smooth"] and not self.smooth_obj_on_main
        fuse = self.pair_fused and on["inter"] and Vo <= 4096 and not self.inter_min
        nn_fused = fuse and not on["con"]
        # with the contact term the FULL search (nearest object vertex of every hand vertex) is the launch's first block
        # range instead, and the contact launches follow it: one launch less on the hand-side chain of the step-2 sets
        nn_full_fused = fuse and on["con"] and self.nn_full_fused
        ht_fused = fuse and self.hand_terms_fused and on["smooth"] and on["v2d"]
        ht_args = (P(m.ref_verts2d_hand), float(m.image_size), P(self.U_v2d), self._slot("loss_v2d_hand"), P(self.U_smh),
                   self._slot("loss_smooth_hand"), P(pca) if pri else None, npca,
                   P(m.int_scales_object), P(m.int_scale_object_mean), P(m.int_scales_hand),
                   P(m.int_scale_hand_mean), P(self.U_pca), P(self.U_so), P(self.U_sh), self._slot("loss_pca"))
        if ht_fused:
            pass
        elif on["smooth"] and on["v2d"]:       # the three hand-only reductions in one launch
            ck(L.hm_hand_terms_fwd_clips(P(self.vh), P(m.camintr), 1, ht_args[0], ht_args[1], B, Vh, *ht_args[2:], rws_b, CL,
                                         NS, sb), "hand terms")
        else:
            if pri:
                ck(L.hm_priors_fwd_clips(P(pca), npca, P(m.int_scales_object), P(m.int_scale_object_mean),
                                         P(m.int_scales_hand), P(m.int_scale_hand_mean), P(self.U_pca), P(self.U_so),
                                         P(self.U_sh), self._slot("loss_pca"), C, NS, sb), "priors")
            if on["smooth"]:
                ck(L.hm_smooth_fwd_clips(P(self.vh), B, Vh, 1, P(self.U_smh), self._slot("loss_smooth_hand"), rws_b,
                                         CL, NS, sb), "smooth(hand)")
            if on["v2d"]:
                ck(L.hm_v2d_fwd_clips(P(self.vh), P(m.camintr), 1, P(m.ref_verts2d_hand), float(m.image_size), B, Vh,
                                      P(self.U_v2d), self._slot("loss_v2d_hand"), rws_b, CL, NS, sb), "v2d")
        nn_early = False
        if on["sil"]:
            if not self.side_own_vo:
                side.wait_event(self.ev_sil)     # self.vo: camera-space object vertices from the calling stream
            if self.pairs_after_lines:
                # (the metric-only search - it feeds nothing but the logged hand-object distance, and is the longest
                #  launch of the hand side in a batch - can run before that wait, next to the rasteriser)
                nn_early = self.nn_early and on["inter"] and not on["con"] and Vo <= 4096 and not self.inter_min
                if nn_early:
                    ck(L.hm_nn_fwd_rigid_clips(P(self.vh), P(self.vo_b), B, Vh, Vo, None, None, self._slot("handobj_maxdist"),
                                               rws_b, CL, NS, P(self.obj_order),
                                               (P(self.obj_spheres) if self.nn_spheres else None), P(m.rotations_object),
                                               P(m.translations_object), P(m.int_scales_object), P(self.hand_order),
                                               P(self.nn_seed), sb), "nn")
                side.wait_event(self.ev_lines)   # (scheduling only, see the silhouette chain above)
            elif self.pairs_after_raster:
                side.wait_event(self.ev_ras)     # (scheduling only, see __init__)
        # one clip, step-2 sets: the collision term (five SDF launches) and the search / contact / interaction launches
        # both start from the two vertex buffers and feed nothing to each other.  The collision chain stays on this
        # stream; everything else of the hand side - and, behind both, the hand's gradient launches - moves to the third
        it.fuse = fuse
        it.ht_args = ht_args
        it.ht_fused = ht_fused
        it.nn_early = nn_early
        it.nn_full_fused = nn_full_fused
        it.nn_fused = nn_fused
        it.sm_here = sm_here

    def _issue_pair_terms(self, it):
        """B: collision, nearest-vertex search, contact, interaction (centroid or min) - on the side stream, or split over two"""
        (m, L, P, ck, B, Vo, Vh, c, on, w, CL, NS, side, sb, rws_b, cctx, use_aux, fuse, ht_args, ht_fused, nn_early, nn_full_fused, nn_fused, sm_here) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.Vh, it.c, it.on, it.w, it.CL, it.NS, it.side, it.sb, it.rws_b, it.cctx, it.use_aux, it.fuse, it.ht_args, it.ht_fused, it.nn_early, it.nn_full_fused, it.nn_fused, it.sm_here)
        # stream, which joins the calling stream (the HIP graph runtime crashes at replay when a forked stream rejoins
        # the SIDE stream: measured twice; forks that rejoin the origin are fine)
        split = on["col"] and self.col_on_aux and not use_aux
        side2, sb2 = (self.aux, self.aux.cuda_stream) if split else (side, sb)
        if on["col"]:
            if split:
                self.ev_hand.record(side)        # both vertex buffers exist on this stream from here on
            ck(L.hm_collision_fwd_clips(P(self.vh), P(cctx.f0), Vh, cctx.f0.shape[0], P(self.vo_b), P(cctx.f1), Vo,
                                        cctx.f1.shape[0], B, c.SDF_SCALE_FACTOR, P(self.U_colh), P(self.U_colo),
                                        self._slot("loss_collision"), P(cctx.ws), CL, NS, sb), "collision")
            if split:
                self.ev_col.record(side)
                side2.wait_event(self.ev_hand)
        if sm_here and not fuse:
            ck(L.hm_smooth_fwd_clips(P(self.vo_b), B, Vo, 1, P(self.U_smo), self._slot("loss_smooth_obj"), rws_b, CL, NS,
                                     sb2), "smooth(obj)")
        def search_and_contact(stream_obj, rws):
            sx = stream_obj.cuda_stream
            if (on["con"] or on["inter"]) and not nn_fused and not nn_full_fused and not nn_early:
                # (without the contact term only the logged distance is needed: metric-only search - its group table
                #  covers 4096 object vertices, larger meshes take the full search for the same number)
                full = on["con"] or Vo > 4096 or self.inter_min      # ('min' names the closest PAIR: indices needed)
                ck(L.hm_nn_fwd_rigid_clips(P(self.vh), P(self.vo_b), B, Vh, Vo, P(self.nn_idx) if full else None,
                                           P(self.nn_d2) if full else None, self._slot("handobj_maxdist"), rws, CL, NS,
                                           P(self.obj_order), (P(self.obj_spheres) if self.nn_spheres else None), P(m.rotations_object),
                                           P(m.translations_object), P(m.int_scales_object), P(self.hand_order),
                                           None if full else P(self.nn_seed), sx), "nn")
            if on["con"]:
                ck(L.hm_contact_fwd_clips(P(self.vh), P(self.vo_b), P(self.nn_idx), B, Vh, Vo, c.COLLISION_THRESH,
                                          P(self.U_conh), P(self.U_cono), self._slot("loss_contact"), rws, CL, NS, sx),
                   "contact")
        # (the search on a third stream at one clip / step 1: +1 %; -9 % on an 8-clip batch.  Search + contact as a third
        #  branch next to the collision term: the HIP graph runtime crashes at replay when two side branches wait for
        #  each other's events.  Capturing the hand-side forward kernels BEFORE the silhouette chain: -38 %.)
        if not nn_full_fused:
            search_and_contact(side2, rws_b)
        if fuse:
            ck(L.hm_pair_terms_fwd_clips(P(self.vh), P(self.vo_b), P(m.camintr), B, Vh, Vo,
                                         self._slot("handobj_maxdist") if (nn_fused or nn_full_fused) else None,
                                         P(self.obj_order), rws_b,
                                         c.INTERACTION_BBOX_EXPANSION, float(c.INTERACTION_Z_THRESH), P(self.rec),
                                         self._slot("loss_inter"), P(self.reduce_ws_c.buf),
                                         P(self.U_smo) if sm_here else None,
                                         self._slot("loss_smooth_obj") if sm_here else None, P(self.reduce_ws_d.buf),
                                         *(ht_args if ht_fused else (None, 0.0, None, None, None, None, None, 0, None, None,
                                                                     None, None, None, None, None, None)),
                                         P(self.reduce_ws_e.buf), (P(self.obj_spheres) if self.nn_spheres else None), P(m.rotations_object),
                                         P(m.translations_object), P(m.int_scales_object), P(self.hand_order),
                                         P(self.nn_idx) if nn_full_fused else None, P(self.nn_d2) if nn_full_fused else None,
                                         P(self.nn_seed), CL, NS, sb2),
               "pair terms")
            if nn_full_fused:
                search_and_contact(side2, rws_b)          # (the contact launches only: the search ran above)
        elif on["inter"]:
            ck(L.hm_inter_fwd_clips(P(self.vh), P(self.vo_b), P(m.camintr), B, Vh, Vo, c.INTERACTION_BBOX_EXPANSION,
                                    float(c.INTERACTION_Z_THRESH), P(self.rec),
                                    P(self.tmp_inter) if self.inter_min else self._slot("loss_inter"), rws_b, CL,
                                    NS, sb2), "inter")
        if on["inter"] and self.inter_min:
            with torch.cuda.stream(side2):
                # inter_type "min" (losses.py:219-221): on the frames the gate lets through (rec[:, 0], same gate as the
                # centroid form) the smallest squared vertex distance; the search names the pair, the term and its
                # gradient live on the two vertices (hand: rigid pose only - the mesh-detached twin; object: only with a free
                # scale).  A handful of small device ops, same expressions as Losses.compute_interaction_loss.
                flags = (self.rec[:, 0] != 0).float()
                i_star = self.nn_d2.argmin(1)
                j_star = self.nn_idx.gather(1, i_star[:, None]).long()[:, 0]
                diff = self.vh[self.rows, i_star] - self.vo_b[self.rows, j_star]
                self.vals[0, self.SLOTS.index("loss_inter")] = ((diff * diff).sum(1) * flags).sum()
                pull = (2.0 * w["loss_inter"]) * diff * flags[:, None]
                self.G_min_h.zero_()
                self.G_min_h[self.rows, i_star] = pull
                if m.optimize_object_scale:
                    self.G_int_o.zero_()
                    self.G_int_o[self.rows, j_star] = -pull
        elif on["inter"] and m.optimize_object_scale:      # the object side of the term reaches the (free) scale: per-vertex form
            ck(L.hm_inter_bwd(P(self.rec), P(self.up_inter), B, Vh, Vo, None, P(self.G_int_o), sb2), "inter_bwd")
        it.sb2 = sb2
        it.side2 = side2
        it.split = split

    def _issue_depth_terms(self, it):
        """B: ordinal depth term - hand render, pair term per clip, the two depth-map backward passes"""
        (m, L, P, ck, B, Vo, Vh, on, CL, NS, C, sb2, side2) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.Vh, it.on, it.CL, it.NS, it.C, it.sb2, it.side2)
        if on["depth"]:
            ctx_o, ctx_h, m_o, m_h = self.dctx
            Sd, K = ctx_o.S, P(m.camintr)
            self._depth_render(self.vh, ctx_h, Vh, self.d_sil_h, self.d_dep_h, sb2)
            if on["sil"]:
                side2.wait_event(self.ev_dep)        # the object's depth image, from the calling stream
            else:
                self._depth_render(self.vo, ctx_o, Vo, self.d_sil_o, self.d_dep_o, sb2)
            rw_bytes = L.hm_reduce_workspace_bytes()
            for ci in range(C):           # per clip: the term normalises over the clip's own pairs and mask counts
                fr = slice(ci * CL, (ci + 1) * CL)
                args = (P(self.d_dep_o[fr]), P(self.d_dep_h[fr]), P(self.d_sil_o[fr]), P(self.d_sil_h[fr]), P(m_o[fr]),
                        P(m_h[fr]), CL, Sd)
                ck(L.hm_ordinal_depth_fwd(*args, P(self.d_part[8 * ci * CL:]), P(self.d_rec[ci]),
                                          self._slot("loss_depth") + 4 * NS * ci,
                                          self.rws_depth.buf.data_ptr() + rw_bytes * ci, sb2), "ordinal depth")
                if self.d_flags is not None:
                    fo = ci * CL * Sd * (Sd // 64)
                    ck(L.hm_ordinal_depth_bwd_flags(*args, P(self.d_rec[ci]), P(self.up_depth), P(self.d_go[fr]), P(self.d_gh[fr]),
                                                    self.d_flags[0].data_ptr() + fo, self.d_flags[1].data_ptr() + fo, sb2),
                       "ordinal depth bwd")
                else:
                    ck(L.hm_ordinal_depth_bwd(*args, P(self.d_rec[ci]), P(self.up_depth), P(self.d_go[fr]), P(self.d_gh[fr]),
                                              sb2), "ordinal depth bwd")
            # the two depth images' backward passes are independent: the hand's stays here, the object's goes to the
            # calling stream (idle between its sweeps and the object's gradient launch) when that stream made the render
            self.ev_dgrad.record(side2)
            for verts, ctx, V_, g, G, li in (((self.vh, ctx_h, Vh, self.d_gh, self.G_dep_h, 1),) if on["sil"] else
                                             ((self.vo, ctx_o, Vo, self.d_go, self.G_dep_o, 0),
                                              (self.vh, ctx_h, Vh, self.d_gh, self.G_dep_h, 1))):
                ck(L.hm_depth_bwd_sparse(P(verts), K, B, V_, ctx.F, Sd, 1.0, P(g), P(ctx.adj_off), P(ctx.adj_items), P(G),
                                         P(self.d_flags[li]) if self.d_flags is not None else None, P(ctx.workspace), sb2),
                   "depth bwd")

    def _issue_hand_backward(self, it):
        """B: join of the split, forward-done / pair-done events, the hand's rigid + MANO backward"""
        (m, L, P, ck, B, Vh, on, w, CL, pca, rot, betas, mtr, sb2, side2, split) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vh, it.on, it.w, it.CL, it.pca, it.rot, it.betas, it.mtr, it.sb2, it.side2, it.split)
        if split:
            side2.wait_event(self.ev_col)    # the collision term's hand gradients, from the side stream
        self.ev_fwd.record(side2)         # every forward loss value of this stream exists now
        if not self.smooth_obj_on_main:
            self._aux_block(it)
        self.ev_pair.record(side2)        # object-side terms of the pair-wise losses are ready
        # hand (full path: MANO + rigid): smooth + v2d + collision + contact, summed with their weights inside the rigid
        # backward; the interaction term reaches the rigid pose only, as one vector per frame (rec[:, 2:5] / Vh)
        tp, tw, tn = _lib.terms([(self.U_smh if on["smooth"] else None, w["loss_smooth_hand"]),
                                 (self.U_v2d if on["v2d"] else None, w["loss_v2d_hand"]),
                                 (self.U_colh if on["col"] else None, w["loss_collision"]),
                                 (self.U_conh if on["con"] else None, w["loss_contact"]),
                                 (self.G_dep_h if on["depth"] else None, 1.0)])       # (already times its weight)
        g_rig = P(self.G_min_h) if (on["inter"] and self.inter_min) else None
        g_frm = (self.rec.data_ptr() + 8) if (on["inter"] and not self.inter_min) else None
        if m.optimize_mano and self.mano_bwd_rigid:
            # the hand's rigid backward inside the MANO backward's launch: one launch less on this chain
            ck(L.hm_mano_bwd_rigid_clips(self.mctx.ptrs, P(pca), self.P, P(rot), P(betas), B,
                                         P(self.U_pca) if on["pca"] else None, w["loss_pca"], P(pca.grad), P(rot.grad),
                                         P(betas.grad), P(mtr.grad), P(self.mano_state), P(self.mctx.workspace(B)),
                                         P(self.vm), P(m.rotations_hand), P(m.int_scales_hand), tp, tw, tn, g_rig, g_frm, 8,
                                         w["loss_inter"] / Vh, P(m.rotations_hand.grad), P(m.translations_hand.grad), CL,
                                         sb2), "mano_bwd + rigid_bwd(hand)")
        else:
            ck(L.hm_rigid_bwd_clips(P(self.vm if m.optimize_mano else m.verts_hand_og), P(m.rotations_hand),
                                    P(m.int_scales_hand), 0, tp, tw, tn, g_rig, g_frm, 8, w["loss_inter"] / Vh, B, Vh,
                                    P(self.G_mesh) if m.optimize_mano else None, P(m.rotations_hand.grad),
                                    P(m.translations_hand.grad), None, P(self.rigid_ws_h), CL, sb2), "rigid_bwd(hand)")
            if m.optimize_mano:
                ck(L.hm_mano_bwd(self.mctx.ptrs, P(pca), self.P, P(rot), P(betas), B, P(self.G_mesh),
                                 P(self.U_pca) if on["pca"] else None, w["loss_pca"], P(pca.grad), P(rot.grad),
                                 P(betas.grad), P(mtr.grad), P(self.mano_state), P(self.mctx.workspace(B)), sb2),
                   "mano_bwd")

    def _issue_object_backward(self, it):
        """A, second half: [object smoothness], [object depth backward], the object's rigid backward with the silhouette gather"""
        (m, L, P, ck, B, Vo, on, w, CL, NS, main, sa, rws_a, sctx) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.on, it.w, it.CL, it.NS, it.main, it.sa, it.rws_a, it.sctx)
        # ---------------- A: object backward: silhouette gradient + smooth + contact [+ interaction with a free scale],
        # summed with their weights inside the rigid backward
        if on["smooth"] and self.smooth_obj_on_main:
            # the object's smoothness term only feeds the object's pose gradients: it rides the silhouette chain (behind the
            # sweeps) instead of lengthening the hand-side chain, which is the longer one at one clip
            if not on["sil"]:
                main.wait_event(self.ev_vo)
            ck(L.hm_smooth_fwd_clips(P(self.vo), B, Vo, 1, P(self.U_smo), self._slot("loss_smooth_obj"), rws_a, CL, NS,
                                     sa), "smooth(obj)")
            self.ev_smo.record(main)
        if self.smooth_obj_on_main:
            self._aux_block(it)
        if on["depth"] and on["sil"]:
            main.wait_event(self.ev_dgrad)       # d loss / d (object's depth image), from the side stream
            ctx_o = self.dctx[0]
            ck(L.hm_depth_bwd_sparse(P(self.vo), P(m.camintr), B, Vo, ctx_o.F, ctx_o.S, 1.0, P(self.d_go), P(ctx_o.adj_off),
                                     P(ctx_o.adj_items), P(self.G_dep_o),
                                     P(self.d_flags[0]) if self.d_flags is not None else None, P(ctx_o.workspace), sa),
               "depth bwd(obj)")
        sc_obj = m.optimize_object_scale
        # the object's smoothness gradient is formed INSIDE the rigid backward from the camera-space vertices the face setup
        # wrote (same floats as the unit gradient of the smoothness launch times its weight): on the step-1 sets this chain
        # then needs nothing from the side stream before the join - one cross-queue edge less on the iteration's tail
        sm_in = on["smooth"] and on["sil"] and Vo <= 24576 and os.environ.get("HOMAN_SMOOTH_IN_RIGID", "1") != "0"
        side_terms = on["con"] or (on["inter"] and sc_obj) or (on["smooth"] and not sm_in) or not on["sil"]
        if side_terms:
            main.wait_event(self.ev_pair)
        tp, tw, tn = _lib.terms([(self.U_smo if (on["smooth"] and not sm_in) else None, w["loss_smooth_obj"]),
                                 (self.U_cono if on["con"] else None, w["loss_contact"]),
                                 (self.G_int_o if (on["inter"] and sc_obj) else None, 1.0),
                                 (self.G_dep_o if on["depth"] else None, 1.0)])
        if on["sil"]:       # the silhouette term is gathered from the sweeps' per-corner gradients inside this launch
            ck(L.hm_rigid_bwd_sil_clips(P(m.verts_object_og), P(m.rotations_object), P(m.int_scales_object), 1, tp, tw, tn,
                                        L.hm_sil_parts(P(sctx.workspace), B, Vo, sctx.F, sctx.S), P(sctx.adj_off),
                                        P(sctx.adj_items), P(self.vo), P(self.sil_K), 1.0, sctx.F, B, Vo,
                                        P(m.rotations_object.grad), P(m.translations_object.grad),
                                        P(self.g_so_part) if sc_obj else None, P(self.rigid_ws_o), CL, sctx.sum_log2q,
                                        P(self.vo) if sm_in else None, w["loss_smooth_obj"], sa),
               "rigid_bwd(obj) + silhouette gather")
        else:
            ck(L.hm_rigid_bwd_clips(P(m.verts_object_og), P(m.rotations_object), P(m.int_scales_object), 1, tp, tw, tn, None,
                                    None, 0, 0.0, B, Vo, None, P(m.rotations_object.grad), P(m.translations_object.grad),
                                    P(self.g_so_part) if sc_obj else None, P(self.rigid_ws_o), CL, sa), "rigid_bwd(obj)")
        it.sc_obj = sc_obj

    def _issue_join(self, it):
        """tail: silhouette reduction (two streams), join, log row, scale gradients"""
        (m, L, P, ck, B, Vo, on, w, CL, NS, C, main, side, sa, sctx, use_aux, log, sc_obj) = \
            (it.m, it.L, it.P, it.ck, it.B, it.Vo, it.on, it.w, it.CL, it.NS, it.C, it.main, it.side, it.sa, it.sctx, it.use_aux, it.log, it.sc_obj)
        if not use_aux:
            # two streams only: the silhouette reduction rides the tail of the silhouette chain (the shorter one at one
            # clip), the log row follows the join
            if on["sil"] and not self.sil_reduce_in_bwd:
                ck(L.hm_sil_reduce_clips(B, Vo, sctx.F, sctx.S, P(m.keep_sum), self._slot("loss_sil_obj"), None,
                                         P(sctx.workspace), CL, NS, sa), "sil_reduce")
        main.wait_stream(side)               # join
        if use_aux or (on["col"] and self.col_on_aux):
            main.wait_stream(self.aux)
        elif log:
            ck(L.hm_log_total_clips(P(self.vals), P(self.weights), len(self.SLOTS), P(self.opt.step_t),
                                    self.max_steps, P(self.log_buf), C, sa), "log")
        if sc_obj:          # per clip: sum of the frames' d loss / d scale + the scale prior's term
            ck(L.hm_sum_small_clips(P(self.g_so_part), CL, 1.0, P(self.U_so) if on["so"] else None, w["loss_scale_obj"],
                                    P(m.int_scales_object.grad), C, sa), "scale grad")
            if self.shared_scale:
                # ONE scalar tied across the clips: its gradient is the sum of the replicas' gradients - local clips
                # here, ranks in _reduce_shared_scale_grad
                ck(L.hm_sum_small_clips(P(m.int_scales_object.grad), C, 1.0, None, 0.0, P(self.g_shared), 1, sa),
                   "shared scale grad")

    def _forward_backward_hands(self, log=False):
        """two hands per frame: homan_amd/fused_hands.py"""
        from .fused_hands import forward_backward_hands
        return forward_backward_hands(self, log)

    def sil_chain_only(self):
        """Measurement helper (tools/bench_sil_kernels.py --chain): just the silhouette chain of an iteration - face setup,
        raster, lines, sweeps - on the current stream, with nothing on any other stream."""
        m, L, P, ck = self.model, self.L, _lib.ptr, _lib.check
        sctx, B, Vo, CL, NS = m.sil_ctx, self.B, self.Vo, self.clip_len, self.NS
        sa = torch.cuda.current_stream().cuda_stream
        ck(L.hm_sil_fwd_clips(P(m.verts_object_og), P(sctx.faces), 0, P(self.sil_K), B, Vo, sctx.F, sctx.S,
                              1.0, self.ops.NMR_NEAR, self.ops.NMR_FAR, P(self.sil_keep), P(self.sil_ref),
                              None, P(self.pooled), None, P(sctx.work_order), None, None, 0, P(m.rotations_object),
                              P(m.translations_object), P(m.int_scales_object), 1, 1, P(sctx.workspace), CL, NS, P(self.vo),
                              sa), "sil_fwd")
        ck(L.hm_sil_bwd_clips(P(self.vo), P(self.sil_K), B, Vo, sctx.F, sctx.S, 1.0, self.sil_eps, 2,
                              P(self.up_sil), None, P(m.keep_sum), P(sctx.adj_off), P(sctx.adj_items), P(sctx.face_order),
                              None, None, P(sctx.workspace), CL, None, NS, sctx.sum_log2q, sa), "sil_bwd")

    def _depth_render(self, verts, ctx, V_, sil, dep, stream_id):
        """depth + silhouette images of one mesh at the full-image camera (reference homan.py:391,406), all frames"""
        m, L, P = self.model, self.L, _lib.ptr
        _lib.check(L.hm_sil_fwd(P(verts), P(ctx.faces), 0, P(m.camintr), self.B, V_, ctx.F, ctx.S, 1.0, self.ops.NMR_NEAR,
                                self.ops.NMR_FAR, None, None, None, P(sil), None, P(ctx.work_order), P(dep), None, 0, None, None,
                                None, 0, self.depth_persistent, P(ctx.workspace), stream_id), "depth render")

    def _adam_log(self):
        if not self.log_in_adam:
            return None
        return (self.vals, self.weights, len(self.SLOTS), self.max_steps, self.log_buf, self.C)

    def _iteration(self):
        if self.graph is not None:
            self.graph.replay()
            if self.shared_scale:
                self._reduce_shared_scale_grad()
                self.graph_b.replay()
        else:
            self.forward_backward(log=not self.log_in_adam)
            if self.shared_scale:
                self._reduce_shared_scale_grad()
                self._spread_shared_scale_grad()
            self.opt.step(zero_grad=False, log=self._adam_log())

    def run(self, steps):
        if self.graph_k is not None:
            while steps >= self.graph_iters:
                self.graph_k.replay()
                steps -= self.graph_iters
        for _ in range(steps):
            self._iteration()

    def loss_evolution(self, steps, clip=None):
        """{name: [float] * steps} of one clip (reference jointopt.py:184-189); a batch of several clips returns the list
        of its clips' dictionaries unless `clip` picks one."""
        torch.cuda.synchronize()
        host = self.log_buf[:steps].cpu().numpy()           # (steps, C, NS)

        def one(ci):
            out = {k: host[:, ci, self.SLOTS.index(k)].astype(np.float64).tolist() for k in self.keys}
            out["loss"] = host[:, ci, len(self.SLOTS)].astype(np.float64).tolist()
            return out
        if clip is not None:
            return one(clip)
        return one(0) if self.C == 1 else [one(ci) for ci in range(self.C)]
