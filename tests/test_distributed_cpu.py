"""World-size-2 gloo tests (CPU) of the batch-sharding helpers used for multi-GPU sampling:
weight broadcast as one flat buffer, contiguous batch slices, ragged all-gather.  The model runs its
autograd (torch-op) path here; what is under test is the sharding logic, which is device-independent."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.distributed import broadcast_model_, sample_sharded, shard_range
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    from model_util import build_model

    model = build_model("cond", 32, seed=42 if rank == 0 else 7 + rank)  # ranks start with different weights
    sent = broadcast_model_(model, src=0)
    ref = build_model("cond", 32, seed=42)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    assert sent == 4 * sum(v.numel() for v in ref.state_dict().values() if v.is_floating_point())

    T = 4
    inp = synthetic_inputs(batch, 32, 32, seed=3)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(T - 2, batch, 1, 32, 32, generator=g)
    du = DiffusionUtils(T, 1e-4, 0.02, "cpu", "linear")
    whole = du.sample_with_host_noise(inp["x"], model, noise, inp["y"], cond_img=inp["cond_img"],
                                      lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    gathered = sample_sharded(du, model, inp["x"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"],
                              gather=True, host_noise=noise)
    local = sample_sharded(du, model, inp["x"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"],
                           gather=False, host_noise=noise)
    lo, hi = shard_range(batch, rank, world)
    assert gathered.shape == whole.shape
    # every image is independent of its batch neighbours: sharded == unsharded up to conv-algorithm rounding
    assert float((gathered - whole).abs().max() / whole.abs().max()) < 1e-5
    assert torch.equal(local, gathered[lo:hi])
    torch.save(gathered, os.path.join(out_dir, f"rank{rank}.pt"))

    # the sampler that draws its own start image (clean-application call surface), guided, sharded the same way
    from diffusionmodelscustom_amd import diffusion_modules as dm
    from diffusionmodelscustom_amd.distributed import sample_sharded_clean
    from model_util import build_clean_model
    cmodel = build_clean_model("clean_class", 32)
    cdu = dm.DiffusionUtils(T, 1e-4, 0.02, "cpu", "linear", img_size=32, data_scaled=True)
    x0 = torch.randn(batch, 1, 32, 32, generator=g)
    whole_c = cdu.sample_with_host_noise(batch, cmodel, 1, noise, start_image=x0, y=inp["y"], cfg_scale=2.0)
    cmodel.eval()
    got_c = sample_sharded_clean(cdu, cmodel, batch, 1, y=inp["y"], cfg_scale=2.0, host_noise=noise, start_image=x0)
    assert got_c.shape == whole_c.shape and float((got_c - whole_c).abs().max() / whole_c.abs().max()) < 1e-5
    dist.barrier()
    dist.destroy_process_group()


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_same_seed(rank, world, port, out_dir):
    """Every rank calls torch.manual_seed(123), as a torchrun script usually does, and all images share one start image
    and one conditioning: the per-step noise is then the only thing that can tell two images apart."""
    _init(rank, world, port)
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd import diffusion_modules as dm
    from diffusionmodelscustom_amd.distributed import sample_sharded, sample_sharded_clean
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    from model_util import build_clean_model, build_model

    torch.manual_seed(123)
    batch, T = 4, 4
    one = synthetic_inputs(1, 32, 32, seed=3)
    rep = {k: v.expand(batch, *v.shape[1:]).contiguous() for k, v in one.items()}
    model = build_model("cond", 32)
    du = DiffusionUtils(T, 1e-4, 0.02, "cpu", "linear")
    out = sample_sharded(du, model, rep["x"], rep["y"], rep["cond_img"], rep["lsm_cond"], rep["topo_cond"])
    assert du.first_image == 0, "the slice offset must not outlive the call"
    for a in range(batch):
        for b in range(a + 1, batch):
            assert float((out[a] - out[b]).abs().max()) > 1e-3, f"images {a} and {b} were given the same noise"
    torch.save(out, os.path.join(out_dir, f"same_seed_rank{rank}.pt"))

    # the sampler that draws its own start images: identically seeded ranks must not start from the same images either
    cmodel = build_clean_model("clean_class", 32)
    cdu = dm.DiffusionUtils(T, 1e-4, 0.02, "cpu", "linear", img_size=32)
    y = torch.zeros(batch, dtype=torch.long)
    outc = sample_sharded_clean(cdu, cmodel, batch, 1, y=y)
    for a in range(batch):
        for b in range(a + 1, batch):
            assert float((outc[a] - outc[b]).abs().max()) > 1e-3, f"clean sampler: images {a} and {b} coincide"

    # fewer images than ranks: the rank with the empty slice still takes part in the gather
    small = sample_sharded(du, model, rep["x"][:1], rep["y"][:1], rep["cond_img"][:1], rep["lsm_cond"][:1],
                           rep["topo_cond"][:1])
    assert small.shape == (1, 1, 32, 32) and torch.isfinite(small).all()
    smallc = sample_sharded_clean(cdu, cmodel, 1, 1, y=y[:1])
    assert smallc.shape == (1, 1, 32, 32)
    torch.save(small, os.path.join(out_dir, f"small_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_identically_seeded_ranks_draw_different_noise(tmp_path):
    mp.spawn(_worker_same_seed, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "same_seed_rank0.pt"), torch.load(tmp_path / "same_seed_rank1.pt")
    assert torch.equal(a, b)
    assert torch.equal(torch.load(tmp_path / "small_rank0.pt"), torch.load(tmp_path / "small_rank1.pt"))


def _worker_c4_split(rank, world, port, out_dir):
    """The 128x128 benchmark configuration's split (global batch 256 -> 128 per rank at world size 2; a node of eight
    gives 32 each) at a 32x32 stand-in size: slices are contiguous, the gathered batch keeps the global order."""
    _init(rank, world, port)
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.distributed import sample_sharded, shard_range
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    from model_util import build_model

    batch, T = 256, 2  # T = 2: one evaluation, the update adds no noise, so single images can be re-evaluated exactly
    inp = synthetic_inputs(batch, 32, 32, seed=5)
    model = build_model("cond", 32)
    du = DiffusionUtils(T, 1e-4, 0.02, "cpu", "linear")
    assert shard_range(batch, rank, world) == (128 * rank, 128 * rank + 128)
    out = sample_sharded(du, model, inp["x"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
    assert out.shape == (batch, 1, 32, 32)
    if rank == 0:
        pick = torch.tensor([0, 127, 128, 255])
        want = du.sample(inp["x"][pick], model, inp["y"][pick], cond_img=inp["cond_img"][pick],
                         lsm_cond=inp["lsm_cond"][pick], topo_cond=inp["topo_cond"][pick])
        assert float((out[pick] - want).abs().max() / want.abs().max()) < 1e-5
    torch.save(out, os.path.join(out_dir, f"c4_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_c4_split_keeps_global_order(tmp_path):
    mp.spawn(_worker_c4_split, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert torch.equal(torch.load(tmp_path / "c4_rank0.pt"), torch.load(tmp_path / "c4_rank1.pt"))


@pytest.mark.parametrize("batch", [4, 3])  # even and ragged split over two ranks
def test_sharded_sampling_world2(tmp_path, batch):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, batch, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "rank0.pt")
    b = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(a, b)  # both ranks hold the same gathered batch


def test_shard_range_partition():
    from diffusionmodelscustom_amd.distributed import shard_range
    for batch in (1, 7, 8, 64, 255):
        for world in (1, 2, 3, 8):
            edges = [shard_range(batch, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
