"""Batch-sharded sampling over the GPUs of one node (one process per GPU, torch.distributed).

The reference has no multi-GPU sampling (its SLURM launcher starts independent replicas,
DDPM_DANRA_conditional_lumi/DDPM_test_lumi__128x128.sh:8-23).  Every image of a batch is independent for
the whole reverse loop (eval-mode BatchNorm, per-sample Instance/LayerNorm and attention), so the batch
is cut into contiguous slices, one per rank, with exactly two collectives outside the loop: a broadcast
of the parameters (one flat fp32 buffer, RCCL over xGMI on the GPU, gloo on the CPU) before it and an
optional all-gather of the results after it.  Nothing is exchanged inside the denoise loop.
"""
import torch
import torch.distributed as dist

__all__ = ["broadcast_model_", "shard_range", "shard_batch", "sample_sharded", "sample_sharded_clean"]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_model_(model: torch.nn.Module, src: int = 0) -> int:
    """Overwrite every floating-point parameter and buffer of ``model`` with rank ``src``'s, using ONE
    broadcast of a flat buffer (66.5 MB for the conditional network).  Returns the bytes sent."""
    rank, world = _world()
    tensors = [v for v in model.state_dict(keep_vars=True).values() if v.is_floating_point()]
    if world == 1 or not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src)
    if rank != src:
        offset = 0
        with torch.no_grad():
            for t in tensors:
                n = t.numel()
                t.copy_(flat[offset:offset + n].reshape(t.shape))
                offset += n
    return flat.numel() * 4


def shard_range(batch: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of a batch for ``rank``; the first ``batch % world`` ranks get one extra."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(rank: int, world: int, *tensors):
    """Slice every non-None tensor along dim 0 with ``shard_range``."""
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
        else:
            lo, hi = shard_range(t.shape[0], rank, world)
            out.append(t[lo:hi])
    return tuple(out)


def sample_sharded(diffusion, model, x, y=None, cond_img=None, lsm_cond=None, topo_cond=None, gather=True,
                   host_noise=None):
    """``diffusion.sample`` on this rank's slice of the global batch.  With ``gather`` every rank returns the
    whole batch in the original order (ragged slices are padded for the all-gather and trimmed after it)."""
    rank, world = _world()
    batch = x.shape[0]
    xs, ys, cs, ls, ts = shard_batch(rank, world, x, y, cond_img, lsm_cond, topo_cond)
    if host_noise is not None:
        lo, hi = shard_range(batch, rank, world)
        local = diffusion.sample_with_host_noise(xs, model, host_noise[:, lo:hi], ys, cond_img=cs, lsm_cond=ls,
                                                 topo_cond=ts)
    else:
        local = diffusion.sample(xs, model, ys, cond_img=cs, lsm_cond=ls, topo_cond=ts)
    if not gather or world == 1:
        return local
    return _gather_ragged(local, batch, world)


def _gather_ragged(local, batch, world):
    widest = -(-batch // world)
    padded = local.new_zeros((widest,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    keep = []
    for r, part in enumerate(parts):
        lo, hi = shard_range(batch, r, world)
        keep.append(part[:hi - lo])
    return torch.cat(keep, dim=0)


def sample_sharded_clean(diffusion, model, n, channels_hr, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
                         cfg_scale=0.0, gather=True, host_noise=None, start_image=None):
    """The same sharding for ``diffusion_modules.DiffusionUtils.sample(n, model, channels_hr, ...)`` (the sampler that
    draws its own start image): this rank samples its slice of the ``n`` images with the matching slices of the
    conditioning tensors.  Start images and noise come from each rank's own generator unless ``start_image`` /
    ``host_noise`` (global tensors, sliced here) are given."""
    rank, world = _world()
    lo, hi = shard_range(n, rank, world)
    ys, cs, ls, ts = shard_batch(rank, world, y, cond_img, lsm_cond, topo_cond)
    if host_noise is not None or start_image is not None:
        local = diffusion.sample_with_host_noise(
            hi - lo, model, channels_hr, None if host_noise is None else host_noise[:, lo:hi],
            start_image=None if start_image is None else start_image[lo:hi], y=ys, cond_img=cs, lsm_cond=ls,
            topo_cond=ts, cfg_scale=cfg_scale)
    else:
        local = diffusion.sample(hi - lo, model, channels_hr, ys, cs, ls, ts, cfg_scale)
    if not gather or world == 1:
        return local
    return _gather_ragged(local, n, world)
