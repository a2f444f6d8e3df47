"""A rank's clips through resident fused steppers: ShardStepper (clips of ANY shapes as concurrently replayed shape groups,
the tied object scale's collective issued once per step for all of them) and ClipFitter (the dataset walk of reference
fit_vid_dataset.py:190-379 on resident steppers)."""
import ctypes
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import lib as _lib
from .homan import HOMan
from .fused import FusedStepper
from .loopcommon import collate_inputs


def _shape_signature(model):
    """what the clips of one clip batch must share (homan_amd.clipbatch): frames, object topology, hands, sizes, options"""
    faces = model.faces_object[0].detach().cpu().numpy()
    return (int(model.translations_object.shape[0]), int(model.verts_object_og.shape[1]), faces.shape[0], hash(faces.tobytes()),
            tuple(model.hand_sides), bool(model.optimize_mano), bool(model.optimize_object_scale), model.hand_proj_mode,
            int(model.image_size), int(model.losses.sil_ctx.size), int(model.mano_pca_pose.shape[1]),
            bool(model.int_scales_hand.requires_grad), model.losses.inter_type, bool(getattr(model, "ordinal_depth", False)))


class ShardStepper:
    """A rank's clips of ANY shapes (a real Core50 shard: every clip its own object mesh, reference homan/datasets/
    core50.py:22-42, and its own length, fit_vid_dataset.py:190): clips that agree in shape are optimised as ONE clip batch
    (one launch per kernel over all of them, FusedStepper on a list), and the batches of the different shapes follow each
    other inside every iteration, each replayed from its own hipGraph.  Every clip keeps its own optimiser and ends up with
    exactly the result of optimising it alone (bit for bit, like the clips of one batch).

    shared_scale (BASELINE cfg5): ONE object scale tied across all clips of all ranks.  The steppers compute their clips'
    gradient sums, this class adds them, issues the rank's ONE all-reduce per iteration (and its one broadcast at the start)
    - so ranks with different numbers of shape groups, or with none, stay in step - and hands the global sum back."""

    def __init__(self, models, loss_weights, lr, max_steps, shared_scale=False, group=None, capture=True):
        from . import dist as hdist
        self.models, self.shared_scale, self.group, self.hdist = list(models), bool(shared_scale), group, hdist
        groups = OrderedDict()
        for i, mdl in enumerate(self.models):
            groups.setdefault(_shape_signature(mdl), []).append(i)
        self.index, self.steppers = [], []                    # per stepper: positions of its clips in `models`
        make = lambda idxs: FusedStepper([self.models[i] for i in idxs], loss_weights, lr, max_steps, capture=capture,
                                         shared_scale=shared_scale, group=group, collectives=False)
        # A shard of ONE shape with the step-1 loss set runs as TWO clip batches side by side (their hipGraphs replayed concurrently,
        # see run): the tails and latency-bound launches of one batch run under the other's heavy kernels - same box, one batch /
        # two: 4 clips 9 073 / 9 988 it/s, 6: 9 333 / 10 148, 8: 9 565 / 10 198, 12: 9 503 / 10 066, 16: 9 626 / 9 971 (four batches:
        # 7 827 at 8 clips, 9 080 at 16: worse; 2 clips: 7 489 / 7 167).  Not with the collision / contact terms (8 clips: 7 687 /
        # 7 583), not with the ordinal depth term (4 clips: 4 757 / 4 089; 8: 5 009 / 4 599 - three rasters per batch) and not
        # with a tied scale (its iteration is two graph halves around a collective).  Results do not depend on batch composition.
        lwf = {k: float(v) for k, v in loss_weights.items()}
        halve = (os.environ.get("HOMAN_SHARD_SPLIT", "1") != "0" and len(groups) == 1 and not shared_scale and
                 not lwf.get("lw_collision", 0) > 0 and not lwf.get("lw_contact", 0) > 0 and not lwf.get("lw_depth", 0) > 0)
        chunks = []
        for idxs in groups.values():
            if halve and len(idxs) >= 4:
                h = (len(idxs) + 1) // 2
                chunks += [idxs[:h], idxs[h:]]
            else:
                chunks.append(idxs)
        for idxs in chunks:
            try:
                built = [(idxs, make(idxs))]
            except NotImplementedError:
                # a configuration the fused loop takes one clip at a time (two hands per frame, inter_type "min"): the clips of
                # the group become groups of their own - they still run side by side (see run)
                if len(idxs) == 1:
                    raise
                built = [([i], make([i])) for i in idxs]
            for ix, st in built:
                self.index.append(ix)
                self.steppers.append(st)
        if self.shared_scale:
            dev = self.models[0].int_scales_object.device if self.models else None
            if dev is None:     # a rank without clips: the collectives of the others, on the device of the group's backend
                import torch.distributed as tdist
                dev = (torch.device("cuda", torch.cuda.current_device())
                       if (tdist.is_initialized() and tdist.get_backend(group) == "nccl") else torch.device("cpu"))
            start = (self.steppers[0].model.int_scales_object.detach()[:1].clone() if self.steppers
                     else torch.zeros(1, device=dev))
            hdist.broadcast_shared_scalar(start, hdist.group_src(group), group)
            with torch.no_grad():
                for st in self.steppers:
                    st.model.int_scales_object.copy_(start.expand_as(st.model.int_scales_object))
            self.total = torch.zeros(1, device=dev)

    def _group_streams(self):
        """one replay stream per shape group: the groups' hipGraphs run CONCURRENTLY.  A one-clip graph is ~90 us of launch
        and edge latency around ~70 us of kernels, and the kernels of different stages of different groups overlap: eight
        one-clip groups side by side reach 7 800 it/s against 6 290 one after the other (8 clips of ONE shape as a batch: 8 900)."""
        if getattr(self, "_streams", None) is None:
            self._streams = [torch.cuda.Stream() for _ in self.steppers]
        return self._streams

    def run(self, steps):
        concurrent = (len(self.steppers) > 1 and all(st.graph is not None for st in self.steppers) and
                      os.environ.get("HOMAN_SHARD_CONCURRENT", "1") != "0")
        cur = torch.cuda.current_stream()
        if concurrent and not self.shared_scale:
            # independent clips, no collective: every group simply replays its `steps` iterations on its own stream
            streams = self._group_streams()
            for s in streams:
                s.wait_stream(cur)
            for _ in range(steps):
                for st, s in zip(self.steppers, streams):
                    with torch.cuda.stream(s):
                        st.graph.replay()
            for s in streams:
                cur.wait_stream(s)
            return
        for _ in range(steps):
            if not self.shared_scale:
                for st in self.steppers:
                    st._iteration()
                continue
            self.total.zero_()
            if concurrent:      # the two halves of the iteration of every group side by side, the collective in between
                streams = self._group_streams()
                for st, s in zip(self.steppers, streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        st.graph.replay()
                for st, s in zip(self.steppers, streams):
                    cur.wait_stream(s)
                    self.total += st.g_shared
                self.hdist.sync_shared_scalar_grad(self.total, self.group)
                for st, s in zip(self.steppers, streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        st.g_shared.copy_(self.total)
                        st.graph_b.replay()
                for s in streams:
                    cur.wait_stream(s)
                continue
            for st in self.steppers:            # forward + backward of every shape group; st.g_shared = sum over its clips
                if st.graph is not None:
                    st.graph.replay()
                else:
                    st.forward_backward(log=True)
                self.total += st.g_shared
            self.hdist.sync_shared_scalar_grad(self.total, self.group)
            for st in self.steppers:
                st.g_shared.copy_(self.total)
                if st.graph_b is not None:
                    st.graph_b.replay()
                else:
                    st._spread_shared_scale_grad()
                    st.opt.step(zero_grad=False)

    def loss_evolution(self, steps):
        """one dictionary per clip, in the order the models were given"""
        out = [None] * len(self.models)
        for st, idxs in zip(self.steppers, self.index):
            evo = st.loss_evolution(steps)
            evo = evo if isinstance(evo, list) else [evo]
            for i, e in zip(idxs, evo):
                out[i] = e
        return out


def _input_signature(kw, image_size, rend_size):
    """what a resident stepper is built for, read off the collated inputs of a clip (cf. _shape_signature of a built model)"""
    faces = np.ascontiguousarray(torch.as_tensor(kw["faces_object"])[0].cpu().numpy())
    return (int(kw["translations_object"].shape[0]), int(kw["verts_object_og"].shape[1]), faces.shape[0], hash(faces.tobytes()),
            tuple(kw["hand_sides"]), int(kw["mano_pca_pose"].shape[1]), tuple(kw["target_masks_object"].shape[1:]),
            tuple(kw["masks_object"].shape[-2:]), int(image_size), int(rend_size))


class ClipFitter:
    """A stream of clips through RESIDENT steppers: the sample loop of reference fit_vid_dataset.py:190-379 - for every clip
    `optimize_hand_object(...)`, then `model.state_dict()` / `get_verts_*` read back - without rebuilding anything for a clip
    whose shapes have been seen before.  Per shape signature (frames, object topology, hands, sizes) ONE set of device buffers,
    workspaces and ONE captured hipGraph stays resident (`max_resident` signatures, least recently used evicted); a new clip
    of a known shape is copied into the static buffers (`FusedStepper.reload`), the graph replayed `num_iterations` times,
    the results copied out.  What a fresh fit spends on building the model, zero-filling ~0.5 GB of workspace, calibrating
    and capturing (about twice a 400-step fit) is paid once per shape, and the process holds a bounded
    number of graphs however many clips it walks.  Results are bit-identical to fresh fits (tests/test_clip_fitter_gpu.py).

    `clips_per_batch` > 1: clips of one shape are fitted that many at a time as one clip batch (one launch per kernel over
    all of them); a last, smaller group of a shape runs through a stepper of its own size.
    fit(clips) -> one result per clip, in order: {"loss_evolution", "state_dict" (Parameters + the buffers
    fit_vid_dataset.py:366-379 / postprocess.py:16-77 read, host tensors), "verts_object", "verts_hand"}.
    `timing` accumulates the seconds spent per stage {collate, build, load, iterations, read_back} and the clip count."""

    READ_BACK = ["translations_object", "rotations_object", "translations_hand", "rotations_hand", "mano_pca_pose", "mano_rot",
                 "mano_trans", "mano_betas", "int_scales_object", "int_scales_hand", "cams_hand"]

    def __init__(self, loss_weights, num_iterations=400, lr=1e-2, clips_per_batch=1, max_resident=4, class_name="default",
                 hand_proj_mode="persp", optimize_mano=True, optimize_mano_beta=True, optimize_object_scale=False,
                 image_size=640, mano_model=None, rend_size=256, ordinal_depth=False):
        self.lw, self.steps, self.lr = dict(loss_weights), int(num_iterations), float(lr)
        self.cpb, self.max_resident = max(1, int(clips_per_batch)), max(1, int(max_resident))
        self.model_kw = dict(class_name=class_name, int_scale_init=1, hand_proj_mode=hand_proj_mode, optimize_mano=optimize_mano,
                             optimize_mano_beta=optimize_mano_beta, optimize_object_scale=optimize_object_scale,
                             image_size=image_size, mano_model=mano_model, rend_size=rend_size, sync_metrics=False,
                             ordinal_depth=ordinal_depth)
        self.resident = OrderedDict()          # (signature, clips) -> FusedStepper
        self._one_by_one = set()          # shape signatures whose clips the fused loop takes one at a time
        self._graph_only = set()          # ... and those it refuses altogether (ortho, free hand scale): the autograd hipGraph
        self.resident_graph = OrderedDict()    # signature -> GraphStepper (one per shape, reloaded per clip)
        self.timing = dict(collate=0.0, build=0.0, load=0.0, iterations=0.0, read_back=0.0, clips=0, built=0, reused=0)

    def _clock(self):
        import time
        torch.cuda.synchronize()
        return time.perf_counter()

    def _inputs(self, clip):
        kw = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
        kw["camintr"] = clip.get("camintr")
        return kw

    def fit(self, clips):
        t0 = self._clock()
        nt = torch.get_num_threads()
        torch.set_num_threads(1)       # (a few hundred small concatenations: waking a 64-thread pool for each costs 1.5 ms)
        try:
            kws = [self._inputs(c) for c in clips]
        finally:
            torch.set_num_threads(nt)
        self.timing["collate"] += self._clock() - t0
        groups = OrderedDict()
        for i, kw in enumerate(kws):
            groups.setdefault(_input_signature(kw, self.model_kw["image_size"], self.model_kw["rend_size"]), []).append(i)
        results = [None] * len(clips)
        for sig, idxs in groups.items():
            for lo in range(0, len(idxs), self.cpb):
                chunk = idxs[lo:lo + self.cpb]
                for i, r in zip(chunk, self._fit_group(sig, [kws[i] for i in chunk])):
                    results[i] = r
        self.timing["clips"] += len(clips)
        return results

    def _fit_group(self, sig, kws):
        key = (sig, len(kws))
        # configurations the fused loop takes one clip at a time (two hands per frame, inter_type="min"): clip by clip through
        # (sig, 1) steppers, like ShardStepper's singleton groups - decided when the shape is first seen
        if len(kws) > 1 and sig in self._one_by_one:
            return [r for kw in kws for r in self._fit_group(sig, [kw])]
        t0 = self._clock()
        if sig in self._graph_only:
            return self._fit_graph(sig, kws[0], t0)
        stepper = self.resident.get(key)
        if stepper is None:
            models = [HOMan(**self.model_kw, **kw) for kw in kws]
            try:
                stepper = FusedStepper(models, self.lw, self.lr, self.steps)
            except NotImplementedError:
                if len(kws) == 1:
                    # a configuration the fused launch sequence does not cover at all (hand_proj_mode="ortho", a free hand
                    # scale): remembered per signature - no second attempt at the fused constructor for the clips that follow
                    self._graph_only.add(sig)
                    return self._fit_graph(sig, kws[0], t0, model=models[0])
                del models
                self._one_by_one.add(sig)
                return [r for kw in kws for r in self._fit_group(sig, [kw])]
            self.resident[key] = stepper
            while len(self.resident) > self.max_resident:
                self.resident.popitem(last=False)         # (its graph stays in lib._KEPT_GRAPHS: a few kilobytes)
            self.timing["build"] += self._clock() - t0
            self.timing["built"] += 1
        else:
            self.resident.move_to_end(key)
            stepper.reload(kws)
            self.timing["load"] += self._clock() - t0
            self.timing["reused"] += 1
        return self._run_and_read(stepper, stepper.model.models)

    def _fit_graph(self, sig, kw, t0, model=None):
        """One clip of a configuration the fused loop refuses: the same iteration through HOMan.forward + autograd in a
        hipGraph (optimize_hand_object's mode="auto" fallback), on ONE resident GraphStepper per shape signature - the next
        clip of the shape is copied into its model (`GraphStepper.reload`), so a dataset walk captures one graph per shape,
        not one per clip."""
        from .jointopt import GraphStepper
        stepper = self.resident_graph.get(sig)
        if stepper is None:
            model = model if model is not None else HOMan(**self.model_kw, **kw)
            stepper = self.resident_graph[sig] = GraphStepper(model, self.lw, self.lr, self.steps)
            while len(self.resident_graph) > self.max_resident:
                self.resident_graph.popitem(last=False)
            self.timing["build"] += self._clock() - t0
            self.timing["built"] += 1
        else:
            self.resident_graph.move_to_end(sig)
            stepper.reload(kw)
            self.timing["load"] += self._clock() - t0
            self.timing["reused"] += 1
        return self._run_and_read(stepper, [stepper.model])

    def _run_and_read(self, stepper, models):
        t1 = self._clock()
        stepper.run(self.steps)
        t2 = self._clock()
        self.timing["iterations"] += t2 - t1
        evo = stepper.loss_evolution(self.steps)
        evo = evo if isinstance(evo, list) else [evo]
        out = []
        with torch.no_grad():
            for one, e in zip(models, evo):
                sd = {k: getattr(one, k).detach().cpu() for k in self.READ_BACK if hasattr(one, k)}
                out.append(dict(loss_evolution=e, state_dict=sd, verts_object=one.get_verts_object()[0].detach().cpu(),
                                verts_hand=one.get_verts_hand()[0].detach().cpu()))
        self.timing["read_back"] += self._clock() - t2
        return out
