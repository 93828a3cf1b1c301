"""Pins the CPU restatement (oracle/cpu_ref.py) to the golden vectors that oracle/make_golden.py
captured from the reference's own code.  CPU only.  Tolerance: 1e-6 * max|reference| (the restatement
uses the same torch CPU ops; most cases are bit-identical)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from model_util import build_clean_model, build_model, state_dict_cpu
from oracle import cpu_ref
from oracle import golden_spec as gs

TOL = 1e-6


@pytest.fixture(scope="module")
def sd_cond():
    return state_dict_cpu(build_model("cond", 64))


@pytest.mark.parametrize("T,sched", gs.SCHEDULE_CASES)
def test_schedule_bit_exact(golden, T, sched):
    g = golden("schedule")
    betas, alphas, alpha_hat = cpu_ref.schedule(T, gs.BETA_MIN, gs.BETA_MAX, sched)
    assert np.array_equal(betas.numpy(), g[f"{sched}_{T}_betas"])
    assert np.array_equal(alphas.numpy(), g[f"{sched}_{T}_alphas"])
    assert np.array_equal(alpha_hat.numpy(), g[f"{sched}_{T}_alpha_hat"])


def test_embeddings_bit_exact(golden):
    g = golden("embeddings")
    t = torch.tensor(g["t"], dtype=torch.long)
    assert np.array_equal(cpu_ref.encoder_pos_encoding(t).numpy(), g["encoder"])
    assert np.array_equal(cpu_ref.decoder_sinusoidal_embedding(t).numpy(), g["decoder"])
    assert np.array_equal(cpu_ref.decoder_sinusoidal_embedding(t[:2], faithful=True).numpy(), g["decoder"][:2])


def test_per_op(golden, sd_cond):
    g = golden("per_op")
    with torch.no_grad():
        for lvl, C, side in gs.ATTN_LEVELS:
            x = gs.randn(100 + lvl, 1, C, side, side)
            out = cpu_ref.image_self_attention(sd_cond, f"encoder.attention_layers.{lvl}", x)
            assert rel_err(out, g[f"attn{lvl}"]) < 2e-6, lvl
        x = gs.randn(110, 1, 4, 64, 64)
        assert rel_err(torch.nn.functional.conv2d(x, sd_cond["encoder.conv1.weight"], None, 2, 3), g["conv1"]) < TOL
        for lvl, cin, side in ((1, 64, 16), (2, 64, 16), (3, 128, 8), (4, 256, 4)):
            x = gs.randn(120 + lvl, 2, cin, side, side)
            out = cpu_ref.resnet_layer(sd_cond, f"encoder.layer{lvl}", x, 1 if lvl == 1 else 2)
            assert rel_err(out, g[f"layer{lvl}"]) < TOL, lvl
        tt = torch.tensor([7], dtype=torch.long)
        for k, (cin, cout, side) in enumerate(((512, 256, 2), (256, 128, 4), (128, 64, 8), (64, 64, 16))):
            f = gs.randn(130 + k, 1, cin, side, side)
            skip = gs.randn(140 + k, 1, cout, 2 * side, 2 * side)
            out = cpu_ref.decoder_block(sd_cond, f"decoder.residual_layers.{k}", f, skip, tt)
            assert rel_err(out, g[f"dec{k}"]) < 2e-6, k
        f = gs.randn(150, 1, 64, 32, 32)
        assert rel_err(cpu_ref.decoder_block(sd_cond, "decoder.final_layer", f, final=True), g["dec_final"]) < TOL


@pytest.mark.parametrize("variant", list(gs.VARIANTS))
def test_forward64(golden, variant, sd_cond):
    g = golden("forward64")
    sd = sd_cond if variant == "cond" else state_dict_cpu(build_model(variant, 64))
    inp = gs.variant_inputs(variant, 2, 64)
    with torch.no_grad():
        for ti in gs.FORWARD_TIMESTEPS:
            t = torch.full((2,), ti, dtype=torch.long)
            out = cpu_ref.unet_forward(sd, inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
            assert rel_err(out, g[f"{variant}_t{ti}"]) < 5e-6, (variant, ti)


def test_forward_mixed_timesteps_and_fmaps(golden, sd_cond):
    g = golden("forward64")
    inp = gs.variant_inputs("cond", 2, 64)
    with torch.no_grad():
        t = torch.tensor([3, 731], dtype=torch.long)
        out = cpu_ref.unet_forward(sd_cond, inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert rel_err(out, g["cond_tmixed"]) < 5e-6
        inp1 = gs.variant_inputs("cond", 1, 64)
        fmaps = cpu_ref.encoder_forward(sd_cond, inp1["x"], torch.full((1,), 500, dtype=torch.long), inp1["y"],
                                        inp1["cond_img"], inp1["lsm_cond"], inp1["topo_cond"])
        for k, fm in enumerate(fmaps):
            assert rel_err(fm, g[f"cond_fmap{k + 1}"]) < 5e-6, k


def test_forward128(golden):
    sd = state_dict_cpu(build_model("cond", 128))
    inp = gs.variant_inputs("cond", 1, 128)
    with torch.no_grad():
        out = cpu_ref.unet_forward(sd, inp["x"], torch.full((1,), 500, dtype=torch.long), inp["y"], inp["cond_img"],
                                   inp["lsm_cond"], inp["topo_cond"])
    assert rel_err(out, golden("forward128")["cond_t500"]) < 5e-6


def test_chain_T50(golden, sd_cond):
    """Full sampling chain with host noise in the reference's draw order (T=50, B=2)."""
    g = golden("chain")
    T = 50
    inp = gs.variant_inputs("cond", 2, 64)
    trace = {"want": set(gs.chain_checkpoints(T))}
    cpu_ref.sample(sd_cond, inp["x"], T, gs.BETA_MIN, gs.BETA_MAX, inp["y"], inp["cond_img"], inp["lsm_cond"],
                   inp["topo_cond"], noise_fn=gs.chain_noise_fn(gs.SEED + 1), trace=trace)
    for i in gs.chain_checkpoints(T):
        assert rel_err(trace[i], g[f"T50_x_after_{i}"]) < 1e-4, i


def test_update_matches_reference_statements():
    betas, alphas, alpha_hat = cpu_ref.schedule(1000, gs.BETA_MIN, gs.BETA_MAX)
    x, eps, z = gs.randn(1, 2, 1, 8, 8), gs.randn(2, 2, 1, 8, 8), gs.randn(3, 2, 1, 8, 8)
    for i in (999, 500, 1):
        c1, c2, c3 = cpu_ref.update_coefficients(betas, alphas, alpha_hat, i)
        t = torch.full((2,), i, dtype=torch.long)
        a = alphas[t][:, None, None, None]
        ah = alpha_hat[t][:, None, None, None]
        b = betas[t][:, None, None, None]
        want = (1 / torch.sqrt(a)) * (x - ((1 - a) / torch.sqrt(1 - ah)) * eps)
        want = want + torch.sqrt(b) * z
        assert torch.equal(cpu_ref.ddpm_update(x, eps, z, c1, c2, c3), want)


@pytest.mark.parametrize("sched", ["linear", "cosine"])
def test_noise_image_bit_exact(golden, sched):
    """Forward process (used by the training step): the restatement reproduces the reference's noiseImage given its own draw."""
    g = golden("noise_image")
    sc = golden("schedule")
    alpha_hat = torch.from_numpy(sc[f"{sched}_{gs.NOISE_IMAGE_T}_alpha_hat"])
    x = gs.randn(700, *gs.NOISE_IMAGE_SHAPE)
    t = torch.from_numpy(g["t"])
    got = cpu_ref.noise_image(x, t, alpha_hat, torch.from_numpy(g[f"{sched}_noise"]))
    assert np.array_equal(got.numpy(), g[f"{sched}_noisy"])


# ---------------------------------------------------------------- DDPM_clean_application variant (clean.npz)
@pytest.fixture(scope="module")
def sd_clean():
    return {v: state_dict_cpu(build_clean_model(v, 64)) for v in gs.CLEAN_VARIANTS}


@pytest.mark.parametrize("T", gs.CLEAN_SCHEDULE_T)
def test_clean_cosine_schedule_bit_exact(golden, T):
    g = golden("clean")
    betas = cpu_ref.beta_schedule_clean(T, gs.BETA_MIN, gs.BETA_MAX, "cosine")
    alphas = 1 - betas
    assert np.array_equal(betas.numpy(), g[f"cosine_{T}_betas"])
    assert np.array_equal(alphas.numpy(), g[f"cosine_{T}_alphas"])
    assert np.array_equal(torch.cumprod(alphas, dim=0).numpy(), g[f"cosine_{T}_alpha_hat"])


def test_clean_attention_with_feed_forward_and_forwards(golden, sd_clean):
    g = golden("clean")
    sd = sd_clean["clean_cond"]
    with torch.no_grad():
        for lvl, C, side in gs.CLEAN_ATTN_LEVELS:
            out = cpu_ref.image_self_attention_clean(sd, f"encoder.attention_layers.{lvl}", gs.randn(100 + lvl, 1, C, side, side))
            assert rel_err(out, g[f"attn{lvl}"]) < 2e-6, lvl
        out = cpu_ref.decoder_block(sd, "decoder.residual_layers.1", gs.randn(131, 1, 256, 4, 4), gs.randn(141, 1, 128, 8, 8),
                                    torch.tensor([7]), clean=True)
        assert rel_err(out, g["dec1"]) < 2e-6
        t = torch.full((2,), 500, dtype=torch.long)
        inp = gs.variant_inputs("clean_cond", 2, 64)
        out = cpu_ref.unet_forward(sd, inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], clean=True)
        assert rel_err(out, g["clean_cond_t500"]) < 5e-6
        inp = gs.variant_inputs("clean_class", 2, 64)
        sd = sd_clean["clean_class"]
        assert rel_err(cpu_ref.unet_forward(sd, inp["x"], t, inp["y"], clean=True), g["clean_class_t500"]) < 5e-6
        assert rel_err(cpu_ref.unet_forward(sd, inp["x"], t, clean=True), g["clean_class_t500_nolabel"]) < 5e-6


@pytest.mark.parametrize("name", list(gs.CLEAN_CHAINS))
def test_clean_chains(golden, sd_clean, name):
    """Sampling chains of the clean variant (plain, guided with shrunk noise, guided with a small weight on the cosine
    schedule) against the reference's, same draws."""
    g = golden("clean")
    c = gs.CLEAN_CHAINS[name]
    inp = gs.variant_inputs(c["variant"], c["B"], 64)
    x, fn = gs.clean_chain_draws(c["B"], 64)
    trace = {"want": set(gs.CLEAN_CHAIN_CHECKPOINTS)}
    cpu_ref.sample_clean(sd_clean[c["variant"]], x, c["T"], gs.BETA_MIN, gs.BETA_MAX, inp["y"], inp["cond_img"],
                         inp["lsm_cond"], inp["topo_cond"], scheduler=c["scheduler"], cfg_scale=c["cfg_scale"],
                         data_scaled=c["data_scaled"], noise_fn=fn, trace=trace)
    for i in gs.CLEAN_CHAIN_CHECKPOINTS:
        assert rel_err(trace[i], g[f"chain_{name}_x_after_{i}"]) < 1e-4, i


@pytest.mark.parametrize("variant,heads", gs.HEAD_CASES)
def test_other_head_counts(golden, variant, heads):
    """1, 2 and 8 attention heads (the clean application's configuration default is 1): restatement and host modules
    against the reference's forwards."""
    clean = variant in gs.CLEAN_VARIANTS
    g = golden("heads_clean" if clean else "heads_danra")
    model = (build_clean_model if clean else build_model)(variant, 64, n_heads=heads)
    inp = gs.variant_inputs(variant, 2, 64)
    t = torch.full((2,), 500, dtype=torch.long)
    with torch.no_grad():
        out = cpu_ref.unet_forward(state_dict_cpu(model), inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"],
                                   inp["topo_cond"], clean=clean, n_heads=heads)
        assert rel_err(out, g[f"{variant}_heads{heads}_t500"]) < 5e-6
        out = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert rel_err(out, g[f"{variant}_heads{heads}_t500"]) < 5e-6
