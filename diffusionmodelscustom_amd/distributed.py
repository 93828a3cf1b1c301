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


class _slice_of_global_batch:
    """While this rank samples images [lo, hi) of a global batch.  Every identically seeded rank first draws ONE number
    from torch's default generator, the number an unsharded ``sample()`` would draw as its Philox seed at this point.
    The in-kernel (Philox) draws of the slice are then those of the global elements: that seed (unless the caller set
    ``noise_seed``) with ``diffusion.first_image = lo``, so the slices get exactly the noise the unsliced batch would
    get on one device.  torch's generators - which the generic step loop draws its per-step noise from - are forked
    and reseeded from that number + lo; without this, ranks started with one ``torch.manual_seed`` would all add the
    same noise to their slices."""

    def __init__(self, diffusion, lo):
        self.diffusion, self.lo = diffusion, lo
        self.fork = None

    def __enter__(self):
        base = int(torch.randint(0, 2 ** 62, (1,)).item())  # advances every rank's generator by the same amount
        # fork only the generator of the device this sampler works on (none for a CPU run: asking for the current
        # CUDA device would create a GPU context in a gloo-only process)
        dev = torch.device(self.diffusion.device)
        devices = [dev.index if dev.index is not None else torch.cuda.current_device()] if dev.type == "cuda" else []
        self.fork = torch.random.fork_rng(devices=devices)
        self.fork.__enter__()
        self.saved = (getattr(self.diffusion, "first_image", 0), getattr(self.diffusion, "noise_seed", None))
        try:
            torch.manual_seed((base + 0x9E3779B97F4A7C15 * (self.lo + 1)) % (2 ** 63))
            self.diffusion.first_image = self.lo
            if self.saved[1] is None:
                self.diffusion.noise_seed = base
        except BaseException:
            self.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *exc):
        self.diffusion.first_image, self.diffusion.noise_seed = self.saved
        self.fork.__exit__(*exc)
        return False


def sample_sharded(diffusion, model, x, y=None, cond_img=None, lsm_cond=None, topo_cond=None, gather=True,
                   host_noise=None):
    """``diffusion.sample`` on this rank's slice of the global batch.  With ``gather`` every rank returns the
    whole batch in the original order (ragged slices are padded for the all-gather and trimmed after it).  A rank
    whose slice is empty (batch smaller than the world) samples nothing and still takes part in the gather."""
    rank, world = _world()
    batch = x.shape[0]
    lo, hi = shard_range(batch, rank, world)
    xs, ys, cs, ls, ts = shard_batch(rank, world, x, y, cond_img, lsm_cond, topo_cond)
    if hi == lo:
        local = xs.detach().clone()
    elif host_noise is not None:
        local = diffusion.sample_with_host_noise(xs, model, host_noise[:, lo:hi], ys, cond_img=cs, lsm_cond=ls,
                                                 topo_cond=ts)
    else:
        with _slice_of_global_batch(diffusion, lo):
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
    conditioning tensors.  Without ``start_image`` every rank draws the start images of the WHOLE batch on the host, as
    the unsharded sampler does, and keeps its slice (identically seeded ranks thus start from different images, and from
    the same images the unsharded call would use); per-step noise as in ``sample_sharded`` unless ``host_noise`` (a
    global tensor, sliced here) is given."""
    rank, world = _world()
    lo, hi = shard_range(n, rank, world)
    ys, cs, ls, ts = shard_batch(rank, world, y, cond_img, lsm_cond, topo_cond)
    if start_image is None:
        start_image = torch.randn((n, channels_hr, diffusion.img_size, diffusion.img_size))
    if hi == lo:
        local = start_image[lo:hi].to(diffusion.device)
    elif host_noise is not None:
        local = diffusion.sample_with_host_noise(hi - lo, model, channels_hr, host_noise[:, lo:hi],
                                                 start_image=start_image[lo:hi], y=ys, cond_img=cs, lsm_cond=ls,
                                                 topo_cond=ts, cfg_scale=cfg_scale)
    else:
        with _slice_of_global_batch(diffusion, lo):
            local = diffusion.sample_with_host_noise(hi - lo, model, channels_hr, None, start_image=start_image[lo:hi],
                                                     y=ys, cond_img=cs, lsm_cond=ls, topo_cond=ts, cfg_scale=cfg_scale)
    if not gather or world == 1:
        return local
    return _gather_ragged(local, n, world)
