"""Multi-GPU execution of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards by CLIP: clips share nothing -- separate HOMan instance, optimiser and targets -- and the
frames of one clip are coupled by the smoothness loss and the per-clip normalisers, so a clip is never split.
  * BASELINE cfg4: no data-path collective at all (`shard_clips` + one optimiser per shard).
  * BASELINE cfg5: ONE shared object-scale scalar across all clips (an extension; the reference's scale is per clip,
    homan/homan.py:121-130): every step each rank all-reduces (sum) the 4-byte gradient of that scalar, then applies
    the identical Adam update, so the replicas of the scalar stay bit-identical.  The message is latency-bound
    (4 bytes); link bandwidth and ring-vs-tree are irrelevant.
The helpers below are device-agnostic (the CPU/gloo tests drive them with world_size 2).
"""
import torch
import torch.distributed as dist


def shard_clips(num_clips, rank, world_size):
    """Contiguous block of clip indices owned by `rank` (64 clips / 8 GPUs -> 8 each, BASELINE cfg4/5)."""
    per = (num_clips + world_size - 1) // world_size
    lo = min(rank * per, num_clips)
    return list(range(lo, min(lo + per, num_clips)))


def sync_shared_scalar_grad(grad, group=None):
    """Sum the gradient of a shared scalar over all ranks, in place (one all-reduce of 1 fp32 per step)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


def broadcast_shared_scalar(param, src=0, group=None):
    """Make every rank start from rank `src`'s value of the shared scalar."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(param.data, src=src, group=group)
    return param


def max_over_ranks(seconds, device=None, group=None):
    """Timing convention of bench.py: the job takes as long as its slowest rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def optimize_clips_shared_scale(models, optimizers, loss_weights, num_iterations, scale_name="int_scales_object",
                                group=None):
    """Step-2 style loop over this rank's clips with ONE object scale shared by every clip of every rank.

    models / optimizers: this rank's per-clip models (built with optimize_object_scale=True) and their Adam
    optimisers (each owning its model's copy of the scalar).  Per step: forward/backward of every local clip, local
    sum of d loss / d scale, one all-reduce, the summed gradient is written to every local copy, optimisers step.
    All copies see the same gradient sequence from the same start value, hence stay identical."""
    for m in models:
        broadcast_shared_scalar(getattr(m, scale_name), 0, group)
    history = []
    for _ in range(num_iterations):
        local = None
        totals = []
        for model, opt in zip(models, optimizers):
            opt.zero_grad()
            loss_dict, _ = model(loss_weights=loss_weights)
            total = sum(loss_dict[k] * loss_weights[k.replace("loss", "lw")] for k in loss_dict)
            total.sum().backward()
            g = getattr(model, scale_name).grad
            local = g.detach().clone() if local is None else local + g.detach()
            totals.append(float(total.detach().sum()))
        sync_shared_scalar_grad(local, group)
        for model, opt in zip(models, optimizers):
            getattr(model, scale_name).grad.copy_(local)
            opt.step()
        history.append(totals)
    return history
