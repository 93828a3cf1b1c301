"""World-size-2 run of the batch-sharded sampler with BOTH ranks on the one GPU of the test box (gloo for the
process group, the HIP path for the sampling): every rank seeds torch alike, as a torchrun script does, and the slices
must carry exactly the in-kernel (Philox) noise the unsharded call on one device gives the same images."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED, BATCH, T, SIZE = 123, 6, 5, 32


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(dev):
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    return {k: v.to(dev) for k, v in synthetic_inputs(BATCH, SIZE, SIZE, seed=3).items()}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.distributed import broadcast_model_, sample_sharded, shard_range
    from model_util import build_model

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = build_model("cond", SIZE, seed=42 if rank == 0 else 9 + rank).to(dev)
    broadcast_model_(model, src=0)
    inp = _inputs(dev)
    du = DiffusionUtils(T, 1e-4, 0.02, dev, "linear")
    assert du.noise_seed is None

    torch.manual_seed(SEED)
    local = sample_sharded(du, model, inp["x"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], gather=False)
    assert du.first_image == 0 and du.noise_seed is None, "the slice's seed and offset must not outlive the call"
    after = int(torch.randint(0, 2 ** 62, (1,)).item())  # where the default generator stands after the call

    # the same slice, seed and offset spelt out: bit-identical
    torch.manual_seed(SEED)
    base = int(torch.randint(0, 2 ** 62, (1,)).item())
    assert after == int(torch.randint(0, 2 ** 62, (1,)).item()), "the sharded call must consume exactly one draw"
    lo, hi = shard_range(BATCH, rank, world)
    engine = model.hip_engine(dev)
    want = engine.sample(inp["x"][lo:hi].clone(), inp["y"][lo:hi], inp["cond_img"][lo:hi], inp["lsm_cond"][lo:hi],
                         inp["topo_cond"][lo:hi], du.betas, du.alphas, du.alpha_hat, seed=base, first_image=lo)
    assert torch.equal(local, want)
    torch.save(local.cpu(), os.path.join(out_dir, f"slice{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_unsharded_call(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    from conftest import rel_err
    from diffusionmodelscustom_amd import DiffusionUtils
    from model_util import build_model
    dev = torch.device("cuda:0")
    model = build_model("cond", SIZE, seed=42).to(dev)
    inp = _inputs(dev)
    du = DiffusionUtils(T, 1e-4, 0.02, dev, "linear")
    torch.manual_seed(SEED)
    whole = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                      topo_cond=inp["topo_cond"]).cpu()
    halves = torch.cat([torch.load(tmp_path / "slice0.pt"), torch.load(tmp_path / "slice1.pt")])
    assert halves.shape == whole.shape
    # same seed, same global element offsets: the noise is the unsharded call's to the bit; an image's arithmetic
    # depends on its batch neighbours only through the order of a few sums (row tiling follows the batch size)
    for b in range(BATCH):
        assert rel_err(halves[b], whole[b]) < 1e-4, f"image {b}"
    # a different noise stream is far outside that band
    torch.manual_seed(SEED + 1)
    other = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                      topo_cond=inp["topo_cond"]).cpu()
    assert rel_err(other, whole) > 1e-2
