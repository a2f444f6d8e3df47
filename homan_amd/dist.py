"""Multi-GPU execution of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards by CLIP: clips share nothing -- separate HOMan instance, optimiser and targets -- and the
frames of one clip are coupled by the smoothness loss and the per-clip normalisers, so a clip is never split.
  * BASELINE cfg4: no data-path collective at all: `shard_clips` gives every rank a contiguous block of clips, which it
    optimises as ONE clip batch (`optimize_clip_shard`: one launch per kernel over all its clips, one Adam state per clip).
  * BASELINE cfg5: ONE object-scale scalar tied across all clips (an extension; the reference's scale is per clip,
    homan/homan.py:121-130).  The tied loss is the sum of the clips' losses (each with its own scale prior), so the
    scalar's gradient is the sum of the clips' gradients: every step each rank sums its local clips and all-reduces
    (sum) that ONE fp32, then every replica takes the identical Adam step and the replicas stay bit-identical.  The
    message is latency-bound (4 bytes); link bandwidth and ring-vs-tree are irrelevant.
    Product path: `FusedStepper(models, ..., shared_scale=True)` (all-reduce on the compute stream between the two
    captured halves of the iteration, no host synchronisation).  `optimize_clips_shared_scale` below is the same
    semantics as a plain autograd loop, device-agnostic (the CPU/gloo tests drive it with world_size 2 and 3).
"""
import torch
import torch.distributed as dist


def shard_clips(num_clips, rank, world_size):
    """Contiguous, balanced block of clip indices owned by `rank`: the first `num_clips % world_size` ranks take one
    clip more (64 clips / 8 GPUs -> 8 each, BASELINE cfg4/5; 9 clips / 8 ranks -> 2,1,1,...)."""
    base, extra = divmod(num_clips, world_size)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


_WARNED_STAGED = False


def _staged(t, group):
    """gloo moves host memory: a device tensor goes through a host copy there (the test / debugging backend that lets
    several ranks share ONE GPU; under nccl = RCCL the collective runs on the device tensor, on the compute stream)"""
    global _WARNED_STAGED
    staged = t.is_cuda and dist.get_backend(group) == "gloo"
    if staged and not _WARNED_STAGED:
        import warnings
        warnings.warn("homan_amd.dist: device tensors in a gloo group are staged through the host (one blocking copy each way "
                      "per collective): fine for tests on one GPU, not a production path - use backend 'nccl' (RCCL)")
        _WARNED_STAGED = True
    return staged


def sync_shared_scalar_grad(grad, group=None):
    """Sum the gradient of a shared scalar over all ranks, in place (one all-reduce of 1 fp32 per step)."""
    if _active(group):
        if _staged(grad, group):
            host = grad.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            grad.copy_(host)
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


def broadcast_shared_scalar(value, src=0, group=None):
    """Make every rank start from rank `src`'s value of the shared scalar (`value`: tensor, updated in place)."""
    if _active(group):
        if _staged(value, group):
            host = value.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            value.copy_(host)
        else:
            dist.broadcast(value, src=src, group=group)
    return value


def max_over_ranks(seconds, device=None, group=None):
    """Timing convention of bench.py: the job takes as long as its slowest rank."""
    if not _active(group):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def group_src(group=None):
    """global rank of the group's first member: the one source every side of a shared-scalar broadcast must name"""
    return dist.get_global_rank(group, 0) if group is not None else 0


def optimize_clip_shard(models, loss_weights, num_iterations, lr=1e-2, shared_scale=False, group=None):
    """This rank's clips on its GPU: clips of equal shape as ONE clip batch (every kernel launched once per iteration over all
    of them, homan_amd.clipbatch), the batches of different shapes - other object meshes, other lengths - one after the other
    inside every iteration (jointopt.ShardStepper), all replayed from hipGraphs.  -> list of loss_evolution dicts, one per
    clip, in the order given; the models hold their optimised parameters.  cfg4: shared_scale=False, no collective.  cfg5:
    shared_scale=True (models built with optimize_object_scale=True)."""
    from .jointopt import ShardStepper
    # (a rank without clips builds a stepper-less ShardStepper: it issues the one broadcast and the per-step all-reduce of
    #  a tied scale with a zero gradient, like every other rank)
    stepper = ShardStepper(list(models), loss_weights, lr, num_iterations, shared_scale=shared_scale, group=group)
    stepper.run(num_iterations)
    return stepper.loss_evolution(num_iterations)


def optimize_clips_shared_scale(models, optimizers, loss_weights, num_iterations, scale_name="int_scales_object",
                                group=None, device=None):
    """Tied-scale loop over this rank's clips as plain autograd (any model with the HOMan surface, CPU or GPU).

    models / optimizers: this rank's per-clip models (built with optimize_object_scale=True) and their Adam optimisers
    (each owning its model's replica of the scalar); either list may be empty on a rank that owns no clip.  Per step:
    forward/backward of every local clip, local sum of d loss / d scale (zero on an empty rank), ONE all-reduce, the
    summed gradient is written to every local replica, optimisers step.  All replicas see the same gradient sequence
    from the same start value, hence stay identical.  Every rank issues exactly one broadcast and `num_iterations`
    all-reduces whatever its number of clips."""
    if device is None:
        device = getattr(models[0], scale_name).device if models else ("cuda" if dist.is_initialized() and
                                                                       dist.get_backend(group) == "nccl" else "cpu")
    start = getattr(models[0], scale_name).detach().clone() if models else torch.zeros(1, device=device)
    broadcast_shared_scalar(start, group_src(group), group)
    with torch.no_grad():
        for m in models:
            getattr(m, scale_name).copy_(start)
    history = []
    for _ in range(num_iterations):
        local = torch.zeros(1, device=device)
        totals = []
        for model, opt in zip(models, optimizers):
            opt.zero_grad()
            loss_dict, _ = model(loss_weights=loss_weights)
            total = sum(loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict)
            total.sum().backward()
            local = local + getattr(model, scale_name).grad.detach().reshape(1)
            totals.append(float(total.detach().sum()))
        sync_shared_scalar_grad(local, group)
        for model, opt in zip(models, optimizers):
            getattr(model, scale_name).grad.copy_(local)
            opt.step()
        history.append(totals)
    return history
