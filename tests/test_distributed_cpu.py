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
