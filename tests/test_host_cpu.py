"""CPU-side checks of the host layer: checkpoint contract, the autograd path's arithmetic, the
DiffusionUtils surface, and that the C-ABI library loads and exports every declared symbol.
No kernel is launched here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN_DIR, ROOT, rel_err
from model_util import build_clean_model, build_model
from diffusionmodelscustom_amd import DiffusionUtils, hip_engine
from diffusionmodelscustom_amd.modules_DANRA_conditional import SinusoidalEmbedding
from oracle import golden_spec as gs


@pytest.fixture(scope="module")
def contract():
    with open(os.path.join(GOLDEN_DIR, "state_dict.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", list(gs.VARIANTS))
def test_state_dict_contract(contract, variant):
    """Keys, registration order, shapes and parameter order equal the reference's (checkpoints drop in)."""
    m = build_model(variant, 64)
    got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    assert got == contract[variant]["state_dict"]
    assert [k for k, _ in m.named_parameters()] == contract[variant]["parameters"]
    assert sum(p.numel() for p in m.parameters()) == contract[variant]["n_parameters"]


def test_xavier_init_hook_sees_convs():
    """training_DANRA_conditional.py:739-753 walks the model with isinstance(m, nn.Conv2d | nn.ConvTranspose2d)."""
    m = build_model("cond", 64)
    convs = [x for x in m.modules() if isinstance(x, (nn.Conv2d, nn.ConvTranspose2d))]
    assert len(convs) == 2 + 16 + 3 + 5 + 5  # stem pair, BasicBlock convs, shortcuts, decoder convT + conv


def test_autograd_path_matches_golden_forward(golden):
    g = golden("forward64")
    m = build_model("cond", 64)
    inp = gs.variant_inputs("cond", 2, 64)
    with torch.no_grad():
        t = torch.full((2,), 500, dtype=torch.long)
        out = m(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
    assert rel_err(out, g["cond_t500"]) < 5e-6


def test_autograd_path_backward():
    m = build_model("uncond", 32)
    m.train()
    x = torch.randn(2, 1, 32, 32)
    out = m(x, torch.tensor([5, 9]), None, None, None, None)
    out.pow(2).mean().backward()
    assert all(p.grad is not None for n, p in m.named_parameters() if "final_layer.time_projection" not in n
               and "sinusiodal" not in n)


def test_sinusoidal_embedding_bit_exact(golden):
    g = golden("embeddings")
    t = torch.tensor(g["t"], dtype=torch.long)
    assert np.array_equal(SinusoidalEmbedding(256)(t).numpy(), g["decoder"])
    m = build_model("cond", 64)
    assert np.array_equal(m.encoder.pos_encoding(t.unsqueeze(-1).float(), 256).numpy(), g["encoder"])


@pytest.mark.parametrize("T,sched", gs.SCHEDULE_CASES)
def test_diffusion_utils_schedule(golden, T, sched):
    g = golden("schedule")
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, "cpu", sched)
    assert np.array_equal(du.betas.numpy(), g[f"{sched}_{T}_betas"])
    assert np.array_equal(du.alphas.numpy(), g[f"{sched}_{T}_alphas"])
    assert np.array_equal(du.alpha_hat.numpy(), g[f"{sched}_{T}_alpha_hat"])


def test_diffusion_utils_surface():
    du = DiffusionUtils(50, 1e-4, 0.02, "cpu", "linear")
    with pytest.raises(AssertionError, match="x must be a 4D tensor"):
        du.sample(torch.zeros(4, 4), build_model("uncond", 32))
    with pytest.raises(AssertionError, match="scheduler must be linear or cosine"):
        DiffusionUtils(50, 1e-4, 0.02, "cpu", "quadratic")
    t = du.sampleTimesteps(16)
    assert t.shape == (16,) and int(t.min()) >= 1 and int(t.max()) < 50
    xt, noise = du.noiseImage(torch.ones(3, 1, 8, 8), torch.tensor([1, 10, 49]))
    assert xt.shape == noise.shape == (3, 1, 8, 8)


def test_generic_sample_loop_on_cpu_matches_golden_chain(golden):
    """CPU tensors take the generic step loop (the reference's own control flow) - T=50 chain."""
    g = golden("chain")
    m = build_model("cond", 64)
    du = DiffusionUtils(50, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear")
    inp = gs.variant_inputs("cond", 2, 64)
    fn = gs.chain_noise_fn(gs.SEED + 1)
    noise = torch.stack([fn(i, inp["x"]) for i in range(49, 1, -1)])
    out = du.sample_with_host_noise(inp["x"], m, noise, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                                    topo_cond=inp["topo_cond"])
    assert rel_err(out, g["T50_x_after_1"]) < 1e-4


def test_skip_shape_assertion():
    m = build_model("uncond", 32)
    with pytest.raises(AssertionError, match="feature maps must be of same shape"):
        m.decoder.residual_layers[0](torch.zeros(1, 512, 1, 1), torch.zeros(1, 256, 3, 3), torch.tensor([1]))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed) and exports each function declared in include/danra_hip.h."""
    header = open(os.path.join(ROOT, "include", "danra_hip.h")).read()
    declared = set(re.findall(r"\b(danra_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(hip_engine.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(hip_engine.library_path())
    for name in declared:
        assert hasattr(lib, name), name


def test_create_rejects_bad_geometry_without_gpu():
    lib = hip_engine.load_library()
    cfg = hip_engine.DanraConfig(1, 1, 1, 4, 256, 1, 3, (ctypes.c_int * 4)(2, 2, 2, 2))
    h = ctypes.c_void_p()
    assert lib.danra_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"n_heads" in lib.danra_last_error()
    cfg.n_heads = 4
    assert lib.danra_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    names = {lib.danra_param_name(h, i).decode() for i in range(lib.danra_num_params(h))}
    assert "encoder.layer2.0.downsample.0.weight" in names and "encoder.label_emb.weight" in names
    assert lib.danra_set_param(h, b"encoder.nope", None, 0) == -5
    # shape checks come before any device work: not packed yet -> DANRA_ESTATE, then the geometry messages
    assert lib.danra_unet_forward(h, None, None, None, None, None, None, None, 1, 64, 64, None) == -2
    assert lib.danra_destroy(h) == 0


def test_cuda_inference_without_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(hip_engine, "_lib", None)
    monkeypatch.setattr(hip_engine, "_LIB_PATH", "/nonexistent/libdanra_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        hip_engine.load_library()


def test_noise_image_cpu_path_matches_golden(golden):
    """DiffusionUtils.noiseImage on CPU tensors is the reference's expression (and draws with torch's generator)."""
    from diffusionmodelscustom_amd import DiffusionUtils
    g = golden("noise_image")
    du = DiffusionUtils(gs.NOISE_IMAGE_T, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear")
    x = gs.randn(700, *gs.NOISE_IMAGE_SHAPE)
    t = torch.from_numpy(g["t"])
    torch.manual_seed(gs.SEED + 7)
    noisy, noise = du.noiseImage(x, t)
    assert np.array_equal(noise.numpy(), g["linear_noise"]) and np.array_equal(noisy.numpy(), g["linear_noisy"])
    noisy2, noise2 = du.noiseImage(x, t, noise=noise)
    assert torch.equal(noisy2, noisy) and noise2 is noise
    with pytest.raises(AssertionError, match="x must be a 4D tensor"):
        du.noiseImage(x[0], t)


def test_checkpoint_round_trip_and_training_step(tmp_path):
    """The reference's checkpoint format and training step on top of these modules:
    torch.save({'network_params', 'optimizer_params'}) as training_DANRA_conditional.py:755-772 writes it, strict
    load into a fresh model, one AdamW step on the noise-prediction loss (training:963-984) that changes the weights."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model = build_model("cond", 32)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    path = tmp_path / "ckpt.pth.tar"
    torch.save({"network_params": model.state_dict(), "optimizer_params": opt.state_dict()}, path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    fresh = build_model("cond", 32, seed=gs.SEED + 1)
    fresh.load_state_dict(ckpt["network_params"])  # strict
    inp = gs.variant_inputs("cond", 2, 32)
    t = torch.tensor([3, 40])
    with torch.no_grad():
        a = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        b = fresh(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
    assert torch.equal(a, b)
    du = DiffusionUtils(100, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear")
    model.train()
    tt = du.sampleTimesteps(2)
    assert tt.min() >= 1 and tt.max() < 100
    noisy, noise = du.noiseImage(inp["x"], tt)
    before = model.decoder.final_layer.conv.weight.detach().clone()
    loss = torch.nn.functional.mse_loss(model(noisy, tt, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]), noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and not torch.equal(before, model.decoder.final_layer.conv.weight)


# ---------------------------------------------------------------- DDPM_clean_application call surface
@pytest.mark.parametrize("variant", list(gs.CLEAN_VARIANTS))
def test_clean_state_dict_contract(variant):
    """Keys, order and shapes of unet.DiffusionNet equal the reference's DDPM_clean_application/src/unet.py network."""
    with open(os.path.join(GOLDEN_DIR, "state_dict_clean.json")) as f:
        contract = json.load(f)[variant]
    m = build_clean_model(variant, 64)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == contract["state_dict"]
    assert [k for k, _ in m.named_parameters()] == contract["parameters"]
    assert sum(p.numel() for p in m.parameters()) == contract["n_parameters"]


def test_clean_modules_match_golden_forward(golden):
    g = golden("clean")
    t = torch.full((2,), 500, dtype=torch.long)
    with torch.no_grad():
        m = build_clean_model("clean_cond", 64)
        inp = gs.variant_inputs("clean_cond", 2, 64)
        out = m(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert rel_err(out, g["clean_cond_t500"]) < 5e-6
        m = build_clean_model("clean_class", 64)
        inp = gs.variant_inputs("clean_class", 2, 64)
        assert rel_err(m(inp["x"], t, inp["y"]), g["clean_class_t500"]) < 5e-6
        assert rel_err(m(inp["x"], t), g["clean_class_t500_nolabel"]) < 5e-6


def test_clean_diffusion_utils_surface_and_guided_chain(golden):
    """diffusion_modules.DiffusionUtils: cosine schedule bit-exact, noiseImage shrink, and the guided chain (shrunk
    noise, cfg_scale 3) through the generic CPU loop against the reference's, same draws; train mode restored."""
    from diffusionmodelscustom_amd import diffusion_modules as dm
    g = golden("clean")
    for T in gs.CLEAN_SCHEDULE_T:
        du = dm.DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, "cpu", "cosine")
        assert np.array_equal(du.betas.numpy(), g[f"cosine_{T}_betas"])
        assert np.array_equal(du.alpha_hat.numpy(), g[f"cosine_{T}_alpha_hat"])
    du = dm.DiffusionUtils()
    assert (du.n_timesteps, du.img_size, du.data_scaled, du.scheduler) == (1000, 64, False, "linear")
    with pytest.raises(AssertionError, match="x must be a 4D tensor but is 3D tensor"):
        du.noiseImage(torch.zeros(1, 8, 8), torch.tensor([1]))
    small = dm.DiffusionUtils(50, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear", img_size=8, data_scaled=True)
    torch.manual_seed(3)
    _, noise = small.noiseImage(torch.zeros(4, 1, 8, 8), torch.tensor([1, 5, 20, 49]))
    torch.manual_seed(3)
    assert torch.equal(noise, torch.randn(4, 1, 8, 8) * 0.005)

    c = gs.CLEAN_CHAINS["guided"]
    m = build_clean_model(c["variant"], 64)
    du = dm.DiffusionUtils(c["T"], gs.BETA_MIN, gs.BETA_MAX, "cpu", c["scheduler"], img_size=64, data_scaled=c["data_scaled"])
    inp = gs.variant_inputs(c["variant"], c["B"], 64)
    x, fn = gs.clean_chain_draws(c["B"], 64)
    noise = torch.stack([fn(i, x) for i in range(c["T"] - 1, 1, -1)])
    out = du.sample_with_host_noise(c["B"], m, 1, noise, start_image=x, y=inp["y"], cfg_scale=c["cfg_scale"])
    assert rel_err(out, g["chain_guided_x_after_1"]) < 1e-4
    assert m.training  # reference diffusion_modules.py:182
    # without the hooks the start image and the draws come from torch's generator, in the reference's order
    torch.manual_seed(gs.SEED + 2)
    m.eval()
    out2 = du.sample(c["B"], m, 1, y=inp["y"], cfg_scale=c["cfg_scale"])
    assert rel_err(out2, g["chain_guided_x_after_1"]) < 1e-4


def test_generation_files_have_the_reference_layout(tmp_path):
    """The six .npz files of a generation run carry the same key, dtype and per-sample shape as the files
    the reference ships (final_generated_samples/, metadata in tests/golden/generation_files.json), and load back."""
    from diffusionmodelscustom_amd import generation_io as gio
    with open(os.path.join(GOLDEN_DIR, "generation_files.json")) as f:
        ref = json.load(f)
    n = 5
    name = gio.save_name("DDPM_conditional_ERA5", "temp", "64x64", "ERA5_cond_lsm_topo_random__simple__4_seasons_"
                         "ValidSplitInTime_9yrs", 100)
    assert name == ref["run"]
    gen = torch.randn(n, 1, 64, 64)
    season = torch.randint(0, 4, (n,))
    point = torch.randint(0, 500, (4, n))
    paths = gio.save_generation_samples(str(tmp_path), name, gen, gen + 1, torch.ones_like(gen), gen * 2, season, point)
    for kind, meta in ref["files"].items():
        assert os.path.basename(paths[kind]) == f"{kind}_samples__{ref['run']}"
        with np.load(paths[kind], allow_pickle=False) as z:
            assert z.files == list(meta)
            shape, dtype = meta["arr_0"]
            got = z["arr_0"]
            assert str(got.dtype) == dtype and got.ndim == len(shape)
            assert list(got.shape) == [n if d == 100 else d for d in shape]
    back = gio.load_generation_samples(str(tmp_path), name)
    assert np.array_equal(back["gen"], gen.numpy()) and np.array_equal(back["season"], season.numpy())
    # an unconditional run saves None for what it does not use; nothing is unpickled on the way back
    gio.save_generation_samples(str(tmp_path), "u.npz", gen)
    back = gio.load_generation_samples(str(tmp_path), "u.npz")
    assert back["cond"] is None and back["season"] is None and np.array_equal(back["gen"], gen.numpy())


def test_dataset_dict_contract_feeds_a_training_epoch():
    """Batches with the reference dataset's dict keys (data_DANRA_conditional.py:513-564) go through one epoch of the
    training step as the reference's pipeline arranges it (training_DANRA_conditional.py:775-984: keys -> positional model
    arguments, sampleTimesteps, noiseImage, MSE on the predicted noise, optimiser step), here restated on this package."""
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.synthetic import SyntheticDANRADataset
    ds = SyntheticDANRADataset(4, 32, sdf_weighted_loss=True)
    item = ds[0]
    assert list(item) == ['img', 'classifier', 'img_cond', 'lsm', 'sdf', 'topo', 'points']
    assert item['img'].shape == (1, 32, 32) and item['classifier'].dtype == torch.int64 and item['points'].shape == (4,)
    assert list(SyntheticDANRADataset(1, 32, conditional_images=False, lsm_topo=False)[0]) == ['img', 'points']
    model = build_model("cond", 32)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6)
    du = DiffusionUtils(100, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear")
    before = model.encoder.conv1.weight.detach().clone()
    total = 0.0
    for samples in torch.utils.data.DataLoader(ds, batch_size=2):
        images = samples['img'].to(torch.float)
        t = du.sampleTimesteps(images.shape[0])
        x_t, noise = du.noiseImage(images, t)
        opt.zero_grad()
        pred = model(x_t, t, samples['classifier'], samples['img_cond'].to(torch.float), samples['lsm'].to(torch.float),
                     samples['topo'].to(torch.float))
        loss = torch.nn.functional.mse_loss(pred, noise)
        loss.backward()
        opt.step()
        total += loss.item()
    assert total == total and total > 0 and not torch.equal(before, model.encoder.conv1.weight)


def test_cpu_training_step_instance_norm_gradients_follow_closed_form():
    """torch's CPU instance_norm backward is wrong for a gradient laid out [H][W][B][C], which is what the attention
    block behind a decoder block's second InstanceNorm sends back; DecoderBlock hands the norm a contiguous gradient.
    Every InstanceNorm of a CPU training step must reproduce the closed form
    dx = (g - mean(g) - xhat * mean(g * xhat)) / sigma evaluated in float64."""
    torch.manual_seed(11)
    model = build_model()
    model.train()
    B = 2
    x, target = torch.randn(B, 1, 64, 64), torch.randn(B, 1, 64, 64)
    planes = [torch.randn(B, 1, 64, 64) for _ in range(3)]
    worst = []

    def watch(mod, inp, out):
        if not torch.is_tensor(out) or out is inp[0]:
            return
        seen = {}
        xin = inp[0]

        def on_dx(dx):
            xd, gd = xin.detach().double(), seen["g"].double()
            mu = xd.mean((2, 3), keepdim=True)
            sigma = (xd.var((2, 3), unbiased=False, keepdim=True) + 1e-5).sqrt()
            xh = (xd - mu) / sigma
            closed = (gd - gd.mean((2, 3), keepdim=True) - xh * (gd * xh).mean((2, 3), keepdim=True)) / sigma
            worst.append(((dx.double() - closed).abs().max() / closed.abs().max()).item())

        xin.register_hook(on_dx)
        out.register_hook(lambda g: seen.__setitem__("g", g.detach().clone()))

    for blk in list(model.decoder.residual_layers) + [model.decoder.final_layer]:
        blk.instance_norm1.register_forward_hook(watch)
        blk.instance_norm2.register_forward_hook(watch)
    pred = model(x, torch.tensor([17, 640]), torch.tensor([1, 3]), *planes)
    torch.nn.functional.mse_loss(pred, target).backward()
    assert len(worst) == 9
    assert max(worst) < 5e-6, worst


def test_eager_train_step_is_the_reference_loop_body_and_graphing_needs_a_gpu():
    """training.eager_train_step = the seven statements of the reference's loop body
    (training_DANRA_conditional.py:608-624); with the draws given it is reproducible.  GraphedTrainStep refuses CPU
    tensors and optimizers that cannot be captured."""
    from diffusionmodelscustom_amd.training import GraphedTrainStep, eager_train_step
    torch.manual_seed(2)
    B = 2
    batch = dict(images=torch.randn(B, 1, 32, 32), t=torch.tensor([5, 150]), noise=torch.randn(B, 1, 32, 32))
    du = DiffusionUtils(200, 1e-4, 0.02)
    results = []
    for _ in range(2):
        model = build_model("uncond", 32)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        loss = eager_train_step(model, opt, du, nn.MSELoss(), **batch)
        results.append((loss.item(), model.decoder.final_layer.conv.weight.detach().clone()))
    assert results[0][0] == results[1][0] and torch.equal(results[0][1], results[1][1])
    model = build_model("uncond", 32)
    model.train()
    x_t, noise = du.noiseImage(batch["images"], batch["t"], batch["noise"])
    by_hand = nn.MSELoss()(model(x_t, batch["t"], None, None, None, None), noise)
    assert abs(by_hand.item() - results[0][0]) <= 1e-6 * abs(by_hand.item())
    with pytest.raises(RuntimeError, match="CUDA"):
        GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-2), du, nn.MSELoss(), batch)
