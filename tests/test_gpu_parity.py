"""GPU parity tests: the HIP path, called through the C ABI, against (a) the golden vectors captured
from the reference's own code and (b) the CPU restatement on the same seeded inputs.

Tolerances (error relative to the tensor's max magnitude, fp32 end to end):
  per-op kernels      1e-5
  one UNet forward    1e-4
  full sampling chain 1e-3
  DDPM update         bit-exact
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from model_util import build_model, state_dict_cpu
from oracle import cpu_ref
from oracle import golden_spec as gs

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5
TOL_FWD = 1e-4
TOL_CHAIN = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cond64(dev):
    from gpu_util import Ops
    model = build_model("cond", 64)
    sd = state_dict_cpu(model)
    model = model.to(dev)
    engine = model.hip_engine(dev)
    return model, sd, Ops(engine)


def test_native_library_is_loaded(cond64):
    import os
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libdanra_hip.so" in maps


def test_time_embeddings_and_projections(cond64, golden, dev):
    _, sd, ops = cond64
    g = golden("embeddings")
    t = torch.tensor(g["t"], dtype=torch.long)
    enc, dec, _ = ops.time_embed(t.to(dev))
    # sin/cos of arguments up to ~1e3: device and host libm differ by a few ulp of the result
    assert np.max(np.abs(enc.cpu().numpy() - g["encoder"])) < 2e-6
    assert np.max(np.abs(dec.cpu().numpy() - g["decoder"])) < 2e-6
    y = torch.tensor([0, 1, 2, 3, 1])
    _, _, tb = ops.time_embed(t.to(dev), y.to(dev))
    emb_e = cpu_ref.encoder_pos_encoding(t) + torch.nn.functional.embedding(y, sd["encoder.label_emb.weight"])
    emb_d = cpu_ref.decoder_sinusoidal_embedding(t)
    want = [cpu_ref.time_projection(sd, f"encoder.time_projection_layers.{i}", emb_e) for i in range(5)]
    want += [cpu_ref.time_projection(sd, f"decoder.residual_layers.{k}.time_projection_layer", emb_d) for k in range(4)]
    assert rel_err(tb.cpu(), torch.cat(want, dim=1)) < TOL_OP


@pytest.mark.parametrize("lvl,C,side", gs.ATTN_LEVELS)
def test_attention_levels(cond64, golden, dev, lvl, C, side):
    _, _, ops = cond64
    x = gs.randn(100 + lvl, 1, C, side, side)
    out = ops.attention(lvl, x.to(dev))
    assert rel_err(out.cpu(), golden("per_op")[f"attn{lvl}"]) < TOL_OP


def test_attention_batch_and_relu(cond64, dev):
    """Batch > 1, the decoder's trailing ReLU, and a 128x128-image shape (L = 4096) against the restatement."""
    _, sd, ops = cond64
    x = gs.randn(7, 3, 64, 32, 32)
    want = torch.relu(cpu_ref.image_self_attention(sd, "decoder.residual_layers.3.attention", x))
    assert rel_err(ops.attention(8, x.to(dev), relu=True).cpu(), want) < TOL_OP
    x = gs.randn(8, 1, 64, 64, 64)
    want = cpu_ref.image_self_attention(sd, "encoder.attention_layers.0", x)
    assert rel_err(ops.attention(0, x.to(dev)).cpu(), want) < TOL_OP
    x = gs.randn(9, 2, 128, 16, 16)  # L = 256, head_dim 32 (level 2 of a 128x128 image)
    want = cpu_ref.image_self_attention(sd, "encoder.attention_layers.2", x)
    assert rel_err(ops.attention(2, x.to(dev)).cpu(), want) < TOL_OP


@pytest.mark.parametrize("index,prefix,C,side,batch", [
    (2, "encoder.attention_layers.2", 128, 8, 5),                 # 64 tokens: one workgroup per (image, head)
    (6, "decoder.residual_layers.1.attention", 128, 8, 3),
    (3, "encoder.attention_layers.3", 256, 4, 6),                 # 16 tokens
    (5, "decoder.residual_layers.0.attention", 256, 4, 2),
    (4, "encoder.attention_layers.4", 512, 2, 8),                 # 4 tokens: QKV projection with one wave per 16 tokens of
    (4, "encoder.attention_layers.4", 512, 2, 4),                 # (four images), attention per image
    (4, "encoder.attention_layers.4", 512, 2, 6),                 # ragged last row tile
])
def test_short_sequence_attention_blocks(cond64, dev, index, prefix, C, side, batch):
    """The attention blocks of the deep levels at batches > 1 against the restatement: LayerNorm + QKV projection +
    attention as one kernel per (image, head) at 64 and 16 tokens (k_attn_block_small), one wave per 16 tokens for the QKV
    and output projections elsewhere (k_proj_small: at 4 tokens a wave's tile holds four images)."""
    _, sd, ops = cond64
    x = gs.randn(300 + index + batch, batch, C, side, side)
    x[0, :, 0, 0] = 0.25          # a token whose channels are all equal: LayerNorm maps it to zero
    if batch > 1:
        x[1] *= 30.0              # images of very different scale side by side
    relu = index >= 5
    want = cpu_ref.image_self_attention(sd, prefix, x)
    if relu:
        want = torch.relu(want)
    got = ops.attention(index, x.to(dev), relu=relu).cpu()
    assert rel_err(got, want) < TOL_OP
    # per image: the error bar above is relative to the whole tensor, the small-scale images must hold it on their own
    for b in range(batch):
        assert rel_err(got[b], want[b]) < 2 * TOL_OP, f"image {b}"


@pytest.mark.parametrize("factor", [3e5, 1e-7])
def test_short_sequence_output_projection_outside_the_split_range(dev, factor):
    """The output projection of the short-sequence blocks (k_proj_small) multiplies rows of attention output; value
    projections scaled by 3e5 push those rows beyond 2^15, scaled by 1e-7 below 2^-8: both take the kernel's fp32
    instructions and must meet the same tolerance."""
    from gpu_util import Ops
    model = build_model("cond", 64)
    with torch.no_grad():
        for prefix, C in (("encoder.attention_layers.2", 128), ("encoder.attention_layers.3", 256), ("encoder.attention_layers.4", 512)):
            mha = model.get_submodule(prefix).attention
            mha.in_proj_weight[2 * C:] *= factor
            mha.in_proj_bias[2 * C:] *= factor
    sd = state_dict_cpu(model)
    ops = Ops(model.to(dev).hip_engine(dev))
    for index, prefix, C, side in ((2, "encoder.attention_layers.2", 128, 8), (3, "encoder.attention_layers.3", 256, 4),
                                   (4, "encoder.attention_layers.4", 512, 2)):
        x = gs.randn(400 + index, 4, C, side, side)
        want = cpu_ref.image_self_attention(sd, prefix, x)
        got = ops.attention(index, x.to(dev)).cpu()
        assert torch.isfinite(got).all()
        assert rel_err(got, want) < TOL_OP, (index, factor)
        if factor < 1:  # the residual dominates: hold the projection's own share to the tolerance as well
            assert rel_err(got - x, want - x) < 10 * TOL_OP, (index, factor)


def test_attention_softmax_is_stable_for_large_logits(cond64, dev):
    """Scores far beyond exp()'s fp32 range must not overflow: the running maximum is subtracted."""
    _, sd, ops = cond64
    x = gs.randn(10, 1, 64, 16, 16)
    x[:, :, 3, 5] *= 400.0  # one token with a huge norm; LayerNorm bounds q/k but the residual path keeps x
    sd2 = dict(sd)
    out = ops.attention(1, x.to(dev)).cpu()
    want = cpu_ref.image_self_attention(sd2, "encoder.attention_layers.1", x)
    assert torch.isfinite(out).all()
    assert rel_err(out, want) < TOL_OP


@pytest.mark.parametrize("index,C,side", [(0, 64, 32), (1, 64, 16), (2, 128, 8), (3, 256, 4), (4, 512, 2)])
def test_attention_propagates_nan_like_torch(cond64, dev, index, C, side):
    """A NaN anywhere in an image's input reaches every output of that image's attention block (LayerNorm spreads it
    over the token, the token's key over every query, as torch's modules:100-107 do), and no other image."""
    _, _, ops = cond64
    x = gs.randn(600 + index, 3, C, side, side)
    clean_out = ops.attention(index, x.to(dev)).cpu()
    x[1, C // 3, side // 2, side - 1] = float("nan")
    out = ops.attention(index, x.to(dev)).cpu()
    assert torch.isnan(out[1]).all(), f"{int((~torch.isnan(out[1])).sum())} values of the poisoned image are not NaN"
    assert torch.equal(out[0], clean_out[0]) and torch.equal(out[2], clean_out[2])


def test_stem_convs(cond64, golden, dev):
    _, _, ops = cond64
    g = golden("per_op")
    x = gs.randn(110, 1, 4, 64, 64).to(dev)
    out = ops.conv1([x[:, i:i + 1].contiguous() for i in range(4)])
    assert rel_err(out.cpu(), g["conv1"]) < TOL_OP
    x = gs.randn(111, 1, 64, 32, 32).to(dev)
    assert rel_err(ops.conv2_bn_relu(x).cpu(), g["conv2_bn_relu"]) < TOL_OP


@pytest.mark.parametrize("lvl,cin,side", ((1, 64, 16), (2, 64, 16), (3, 128, 8), (4, 256, 4)))
def test_resnet_layers(cond64, golden, dev, lvl, cin, side):
    _, _, ops = cond64
    x = gs.randn(120 + lvl, 2, cin, side, side)
    assert rel_err(ops.resnet_layer(lvl, x.to(dev)).cpu(), golden("per_op")[f"layer{lvl}"]) < TOL_OP


@pytest.mark.parametrize("k,cin,cout,side", ((0, 512, 256, 2), (1, 256, 128, 4), (2, 128, 64, 8), (3, 64, 64, 16)))
def test_decoder_blocks(cond64, golden, dev, k, cin, cout, side):
    _, _, ops = cond64
    f = gs.randn(130 + k, 1, cin, side, side)
    skip = gs.randn(140 + k, 1, cout, 2 * side, 2 * side)
    out = ops.decoder_block(k, f.to(dev), skip.to(dev), torch.tensor([7]).to(dev))
    assert rel_err(out.cpu(), golden("per_op")[f"dec{k}"]) < TOL_OP


def test_final_decoder_block(cond64, golden, dev):
    _, _, ops = cond64
    f = gs.randn(150, 1, 64, 32, 32)
    assert rel_err(ops.decoder_block(4, f.to(dev)).cpu(), golden("per_op")["dec_final"]) < TOL_OP


def test_instance_norm(cond64, dev):
    _, _, ops = cond64
    x = 3.0 + 2.0 * gs.randn(11, 2, 64, 32, 32)  # non-zero mean: the two-pass variance must not cancel
    want = torch.nn.functional.instance_norm(x, eps=1e-5)
    assert rel_err(ops.instance_norm(x.to(dev)).cpu(), want) < TOL_OP


def test_ddpm_update_bit_exact(cond64, dev):
    _, _, ops = cond64
    betas, alphas, alpha_hat = cpu_ref.schedule(1000, gs.BETA_MIN, gs.BETA_MAX)
    x, eps, z = gs.randn(1, 2, 1, 64, 64), gs.randn(2, 2, 1, 64, 64), gs.randn(3, 2, 1, 64, 64)
    for i in (999, 500, 2):
        c1, c2, c3 = cpu_ref.update_coefficients(betas, alphas, alpha_hat, i)
        want = cpu_ref.ddpm_update(x, eps, z, c1, c2, c3)
        got = ops.ddpm_update(x.to(dev), eps.to(dev), z.to(dev), float(c1), float(c2), float(c3))
        assert torch.equal(got.cpu(), want), i
    c1, c2, c3 = cpu_ref.update_coefficients(betas, alphas, alpha_hat, 1)
    want = cpu_ref.ddpm_update(x, eps, torch.zeros_like(x), c1, c2, c3)
    assert torch.equal(ops.ddpm_update(x.to(dev), eps.to(dev), None, float(c1), float(c2), float(c3)).cpu(), want)


def test_philox_noise_statistics(cond64, dev):
    import ctypes
    from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream
    _, _, ops = cond64
    n = 1 << 22
    zero = torch.zeros(n, device=dev)
    draws = []
    for step in (5, 6):
        x = torch.zeros(n, device=dev)
        _check(ops.lib, ops.lib.danra_ddpm_update_philox(_ptr(x), _ptr(zero), 1.0, 0.0, 1.0, n, ctypes.c_uint64(42),
                                                         ctypes.c_uint64(step), _stream()), "philox")
        draws.append(x.cpu())
    z = draws[0].double()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.var()) - 1.0) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05            # Gaussian kurtosis
    assert abs(float((draws[0] * draws[1]).mean())) < 3e-3      # steps are independent
    assert abs(float((z[:-1] * z[1:]).mean())) < 3e-3           # neighbours are independent


@pytest.mark.parametrize("variant", list(gs.VARIANTS))
def test_forward64_all_variants(golden, dev, variant, cond64):
    g = golden("forward64")
    model = cond64[0] if variant == "cond" else build_model(variant, 64).to(dev)
    inp = {k: (None if v is None else v.to(dev)) for k, v in gs.variant_inputs(variant, 2, 64).items()}
    with torch.no_grad():
        for ti in gs.FORWARD_TIMESTEPS:
            t = torch.full((2,), ti, dtype=torch.long, device=dev)
            out = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
            assert rel_err(out.cpu(), g[f"{variant}_t{ti}"]) < TOL_FWD, (variant, ti)


def test_forward_mixed_timesteps_and_feature_maps(cond64, golden, dev):
    model, _, _ = cond64
    g = golden("forward64")
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 2, 64).items()}
    with torch.no_grad():
        t = torch.tensor([3, 731], device=dev)
        out = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert rel_err(out.cpu(), g["cond_tmixed"]) < TOL_FWD
        inp1 = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 1, 64).items()}
        model(inp1["x"], torch.full((1,), 500, device=dev), inp1["y"], inp1["cond_img"], inp1["lsm_cond"],
              inp1["topo_cond"])
        for k, fm in enumerate(model.hip_engine(dev).fmaps()):
            assert rel_err(fm.cpu(), g[f"cond_fmap{k + 1}"]) < TOL_FWD, k


def test_forward128(golden, dev):
    model = build_model("cond", 128).to(dev)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 1, 128).items()}
    with torch.no_grad():
        out = model(inp["x"], torch.full((1,), 500, device=dev), inp["y"], inp["cond_img"], inp["lsm_cond"],
                    inp["topo_cond"])
    assert rel_err(out.cpu(), golden("forward128")["cond_t500"]) < TOL_FWD


def test_forward_ragged_batch_and_rectangular_image(dev):
    """Row counts that are not multiples of any tile size, and H != W, against the restatement."""
    model = build_model("lsm_topo", 64)
    sd = state_dict_cpu(model)
    model = model.to(dev)
    g = torch.Generator().manual_seed(5)
    B, H, W = 3, 32, 96
    x, lsm, topo = (torch.randn(B, 1, H, W, generator=g) for _ in range(3))
    t = torch.tensor([1, 400, 999])
    with torch.no_grad():
        want = cpu_ref.unet_forward(sd, x, t, None, None, lsm, topo)
        got = model(x.to(dev), t.to(dev), None, None, lsm.to(dev), topo.to(dev))
    assert rel_err(got.cpu(), want) < TOL_FWD


def test_weights_are_repacked_after_load_state_dict(cond64, dev):
    from diffusionmodelscustom_amd.synthetic import synthetic_state_dict
    model = build_model("uncond", 32).to(dev)
    x = gs.randn(3, 1, 1, 32, 32).to(dev)
    t = torch.tensor([10], device=dev)
    with torch.no_grad():
        a = model(x, t).cpu()
        model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=7))
        b = model(x, t).cpu()
        want = cpu_ref.unet_forward(state_dict_cpu(model), x.cpu(), t.cpu())
    assert rel_err(a, b) > 1e-2
    assert rel_err(b, want) < TOL_FWD


@pytest.mark.parametrize("name", ["T50", "T200"])
def test_sampling_chain_host_noise(cond64, golden, dev, name):
    """DiffusionUtils.sample through danra_sample with the reference's host-noise draw order."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    g = golden("chain")
    T, B = gs.CHAIN_CASES[name]["T"], gs.CHAIN_CASES[name]["B"]
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    cpu_in = gs.variant_inputs("cond", B, 64)
    fn = gs.chain_noise_fn(gs.SEED + 1)
    noise = torch.stack([fn(i, cpu_in["x"]) for i in range(T - 1, 1, -1)]).to(dev)
    inp = {k: v.to(dev) for k, v in cpu_in.items()}
    steps = gs.chain_checkpoints(T)
    engine = model.hip_engine(dev)
    x, trace = engine.sample(inp["x"].clone(), inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], du.betas,
                             du.alphas, du.alpha_hat, noise=noise, trace_steps=steps)
    for k, i in enumerate(steps):
        assert rel_err(trace[k].cpu(), g[f"{name}_x_after_{i}"]) < TOL_CHAIN, (name, i)
    out = du.sample_with_host_noise(inp["x"], model, noise, inp["y"], cond_img=inp["cond_img"],
                                    lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    assert torch.equal(out, x)  # the public entry point is the same deterministic computation
    assert rel_err(out.cpu(), g[f"{name}_x_after_1"]) < TOL_CHAIN


def test_sampling_chain_T1000(cond64, golden, dev):
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    du = DiffusionUtils(1000, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    cpu_in = gs.variant_inputs("cond", 1, 64)
    fn = gs.chain_noise_fn(gs.SEED + 1)
    noise = torch.stack([fn(i, cpu_in["x"]) for i in range(999, 1, -1)]).to(dev)
    inp = {k: v.to(dev) for k, v in cpu_in.items()}
    out = du.sample_with_host_noise(inp["x"], model, noise, inp["y"], cond_img=inp["cond_img"],
                                    lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    assert rel_err(out.cpu(), golden("chain")["T1000_x_after_1"]) < TOL_CHAIN


def test_full_size_batch_properties(cond64, dev):
    """Benchmark-size batch (B = 64, 64x64 conditional): each image's result is independent of its batch
    neighbours (eval-mode network has no cross-sample statistic), checked against B = 1 runs, and
    Philox sampling is deterministic in its seed."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 64, 64, seed=9).items()}
    t = torch.full((64,), 321, device=dev)
    with torch.no_grad():
        full = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        for i in (0, 37, 63):
            one = model(*(inp[k][i:i + 1] for k in ("x",)), t[i:i + 1], inp["y"][i:i + 1], inp["cond_img"][i:i + 1],
                        inp["lsm_cond"][i:i + 1], inp["topo_cond"][i:i + 1])
            assert rel_err(full[i:i + 1].cpu(), one.cpu()) < TOL_OP, i
    du = DiffusionUtils(6, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    du.noise_seed = 1234
    a = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    b = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    du.noise_seed = 1235
    c = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()


def test_two_launches_on_the_whole_chip_are_bit_identical(cond64, dev):
    """Full-chip grids (B = 64: 256 workgroups of the 8x8 stride-2 convolution, the fused attention tail of the last
    decoder block, every split-K reduction) launched twice on the same inputs give the same bits: no result depends on
    which workgroup or wave finishes first.  Checked on the convolution alone through its own entry point, with the
    wave-role kernel and with fp32 instructions, and on the whole evaluation."""
    model, _, ops = cond64
    x = gs.randn(43, 64, 64, 32, 32).to(dev)
    try:
        for mode in (0, 1):
            ops.lib.danra_set_fp32_products(mode)
            a, b = ops.conv2_bn_relu(x), ops.conv2_bn_relu(x)
            assert torch.equal(a, b), f"8x8 stride-2 convolution, fp32 products {mode}"
    finally:
        ops.lib.danra_set_fp32_products(0)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 64, 64, seed=10).items()}
    t = (torch.arange(64, device=dev) * 15) % 999 + 1
    with torch.no_grad():
        outs = [model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all()


def test_graph_replay_equals_enqueued_loop(cond64, dev):
    """The captured-step replay and the kernel-by-kernel loop are the same computation: bit-identical in both
    noise modes, with the trace taken between replays, and across a change of batch size (new capture)."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    engine = model.hip_engine(dev)
    T = 12
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    try:
        for B in (3, 2):
            inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", B, 64, seed=21).items()}
            noise = gs.randn(77, T - 2, B, 1, 64, 64).to(dev)
            res = {}
            for graph in (True, False):
                engine.set_graph_mode(graph)
                xh, tr = engine.sample(inp["x"].clone(), inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"],
                                       du.betas, du.alphas, du.alpha_hat, noise=noise, trace_steps=[T - 1, 5, 1])
                xp = engine.sample(inp["x"].clone(), inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"],
                                   du.betas, du.alphas, du.alpha_hat, seed=99)
                res[graph] = (xh, tr, xp)
            for a, b in zip(res[True], res[False]):
                assert torch.equal(a, b)
            assert torch.equal(res[True][1][2], res[True][0])  # the trace entry of timestep 1 is the final x
    finally:
        engine.set_graph_mode(True)


def test_sampler_loop_equals_its_parts(cond64, dev):
    """danra_sample == forward + update composed by hand through the per-call entry points (Philox stream
    keyed by the timestep, fp32 coefficients as the reference computes them)."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    engine = model.hip_engine(dev)
    ops = cond64[2]
    T, B, seed = 5, 2, 4242
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", B, 64, seed=5).items()}
    got = engine.sample(inp["x"].clone(), inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], du.betas,
                        du.alphas, du.alpha_hat, seed=seed)
    # IEEE fp32 scalar arithmetic (numpy), the way the library's host loop evaluates diffusion:135-157
    betas, alphas, alpha_hat = (v.cpu().numpy() for v in (du.betas, du.alphas, du.alpha_hat))
    one = np.float32(1)
    x = inp["x"].clone()
    for i in range(T - 1, 0, -1):
        t = torch.full((B,), i, device=dev, dtype=torch.long)
        eps = engine.forward(x, t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        c1 = one / np.sqrt(alphas[i])
        c2 = (one - alphas[i]) / np.sqrt(one - alpha_hat[i])
        c3 = np.sqrt(betas[i])
        if i > 1:
            x = ops.ddpm_update_philox(x, eps, float(c1), float(c2), float(c3), seed, i)
        else:
            x = ops.ddpm_update(x, eps, None, float(c1), float(c2), float(c3))
    assert torch.equal(got, x)


def test_slices_of_a_batch_draw_the_noise_of_the_whole_batch(cond64, dev):
    """Two processes that sample the halves of one batch with one seed (batch sharding over GPUs) must not add the same
    noise to both halves: with first_image the in-kernel draws are those of the global elements, so the two slices
    together equal the unsliced call (an image does not depend on its batch neighbours beyond rounding) and identical
    inputs at different global positions give different samples."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    engine = model.hip_engine(dev)
    T, B, seed = 6, 4, 77
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    one = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 1, 64, seed=8).items()}
    inp = {k: v.expand(B, *v.shape[1:]).contiguous() for k, v in one.items()}

    def run(lo, hi, first):
        return engine.sample(inp["x"][lo:hi].clone(), inp["y"][lo:hi], inp["cond_img"][lo:hi], inp["lsm_cond"][lo:hi],
                             inp["topo_cond"][lo:hi], du.betas, du.alphas, du.alpha_hat, seed=seed, first_image=first)
    whole = run(0, B, 0)
    halves = torch.cat([run(0, 2, 0), run(2, 4, 2)])
    assert rel_err(halves.cpu(), whole.cpu()) < TOL_CHAIN
    for a in range(B):
        for b in range(a + 1, B):
            assert float((whole[a] - whole[b]).abs().max()) > 1e-2, (a, b)
    same_stream = run(2, 4, 0)  # without the offset the second half repeats the first half's noise
    assert rel_err(same_stream.cpu(), whole[:2].cpu()) < TOL_CHAIN
    assert torch.equal(run(0, 2, 0), run(0, 2, 0))  # the offset is per call: a later plain call is unchanged


def conv2_reference(sd, x):
    """Encoder.conv2 + bn1 (eval) + ReLU in float64 on the CPU (reference modules:186-191, 269-273)."""
    d = {k: v.double() for k, v in sd.items() if k.startswith("encoder.conv2") or k.startswith("encoder.bn1")}
    y = torch.nn.functional.conv2d(x.double(), d["encoder.conv2.weight"], stride=2, padding=3)
    y = torch.nn.functional.batch_norm(y, d["encoder.bn1.running_mean"], d["encoder.bn1.running_var"],
                                       d["encoder.bn1.weight"], d["encoder.bn1.bias"], training=False, eps=1e-5)
    return torch.relu(y)


@pytest.mark.parametrize("case", ["plain", "huge_value", "non_finite_free_huge_region", "tiny", "tiny_with_zeros", "zeros",
                                  "band_gaussian_3e-3", "band_uniform_2^-8_2^-3", "band_mixed_with_order_one"])
def test_split_operand_convolution_value_ranges(cond64, dev, case):
    """The convolutions multiply on the fp16 matrix cores with both operands split in two fp16 halves while a
    workgroup's staged activations stay below 2^15 and are not uniformly tiny; a workgroup that meets anything else
    multiplies its 64 output rows again with fp32 instructions.  Every case stays within the per-op tolerance of a
    float64 evaluation, judged per image region so that a huge value does not hide the error elsewhere."""
    _, sd, ops = cond64
    x = gs.randn(41, 2, 64, 32, 32)
    if case == "huge_value":
        x[0, 5, 10, 12] = 3.0e6
    elif case == "non_finite_free_huge_region":
        x[1, :, 16:, :] *= 1.0e5
    elif case == "tiny":
        x *= 1.0e-6
    elif case == "tiny_with_zeros":
        x *= 1.0e-5
        x[:, :, ::2] = 0.0
    elif case == "zeros":
        x[0] = 0.0
    elif case == "band_gaussian_3e-3":
        # the band in which the split path is still taken (a tile's largest value is above 2^-8) while the low halves
        # are fp16 subnormals: values are kept to an absolute 2^-25, i.e. to 2^-17 relative at the band's lower end
        x *= 3.0e-3
    elif case == "band_uniform_2^-8_2^-3":
        g = torch.Generator().manual_seed(43)
        mag = torch.exp2(torch.rand(x.shape, generator=g) * 5.0 - 8.0)  # log-uniform in [2^-8, 2^-3)
        x = mag * torch.sign(x)
    elif case == "band_mixed_with_order_one":
        # every tile holds order-one values (so no tile falls back) beside channels that sit deep in the band
        x[:, ::2] *= 4.0e-3
    want = conv2_reference(sd, x)
    got = ops.conv2_bn_relu(x.to(dev)).cpu().double()
    assert torch.isfinite(got).all()
    # regions of comparable magnitude: each image's top and bottom half of the 16x16 output
    for b in range(2):
        for rows in (slice(0, 8), slice(8, 16)):
            w, g = want[b, :, rows], got[b, :, rows]
            scale = max(float(w.abs().max()), 1e-30)
            assert float((g - w).abs().max()) <= TOL_OP * scale, (case, b, rows)


def test_fused_attention_blocks_equal_the_separate_kernels(cond64, dev, monkeypatch):
    """A whole forward at the benchmark's batch (64 images: the fused short-sequence attention kernels and every
    one-trip projection kernel run) against a second engine created with DANRA_FUSED_SMALL_ATTENTION=0, whose blocks
    run their projections as tiled GEMMs and the attention kernel separately."""
    model, _, _ = cond64
    B = 64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", B, 64).items()}
    t = torch.randint(1, 1000, (B,), generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        fused = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).cpu()
    monkeypatch.setenv("DANRA_FUSED_SMALL_ATTENTION", "0")
    other = build_model("cond", 64).to(dev)  # the same seeded weights; its engine is created under the switch
    with torch.no_grad():
        separate = other(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).cpu()
    assert rel_err(fused, separate) < 2e-5
    assert not torch.equal(fused, separate), "the switch changed nothing: one of the two forms did not run"
    for b in range(B):  # no image may be off on its own (a row tile of the 2x2 level holds four of them)
        assert rel_err(fused[b], separate[b]) < 1e-4, f"image {b}"


def test_presplit_keys_and_values_change_no_bit(cond64, dev, monkeypatch):
    """The QKV projections of the 64-channel blocks hand K and V to the attention core already split in fp16 halves
    (with DANRA_PRESPLIT=0 the core splits fp32 K and V itself, tile by tile): the same halves either way, so a whole
    forward at the benchmark's batch - the 1024-token blocks through the row-stationary GEMM, the 256-token ones through
    the one-wave-per-16-tokens kernel - must not move by a bit; likewise single blocks at small batches, where the
    1024-token block takes the second producer."""
    from gpu_util import Ops
    model, _, ops = cond64
    B = 64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", B, 64).items()}
    t = torch.randint(1, 1000, (B,), generator=torch.Generator().manual_seed(6)).to(dev)
    with torch.no_grad():
        pre = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).cpu()
    x0, x1 = gs.randn(31, 2, 64, 32, 32).to(dev), gs.randn(32, 3, 64, 16, 16).to(dev)
    pre0, pre1 = ops.attention(0, x0).cpu(), ops.attention(8, x0, relu=True).cpu()
    pre2 = ops.attention(1, x1).cpu()
    monkeypatch.setenv("DANRA_PRESPLIT", "0")
    other = build_model("cond", 64).to(dev)  # the same seeded weights; its engine is created under the switch
    with torch.no_grad():
        plain = other(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).cpu()
    assert torch.equal(pre, plain)
    ops2 = Ops(other.hip_engine(dev))
    assert torch.equal(ops2.attention(0, x0).cpu(), pre0)
    assert torch.equal(ops2.attention(8, x0, relu=True).cpu(), pre1)
    assert torch.equal(ops2.attention(1, x1).cpu(), pre2)


@pytest.mark.parametrize("index,prefix,side,batch", [(0, "encoder.attention_layers.0", 32, 32),    # 32768 tokens: the row-stationary GEMM
                                                     (0, "encoder.attention_layers.0", 32, 2),     # one wave per 16 tokens
                                                     (1, "encoder.attention_layers.1", 16, 5)])
@pytest.mark.parametrize("case", ["huge_values", "tiny_values", "huge_keys_and_values", "nonfinite_value"])
def test_presplit_producers_flag_what_leaves_the_split_range(dev, monkeypatch, index, prefix, side, batch, case):
    """Value and key projections scaled out of the range in which the fp16 halves hold 22 bits: the projection kernel
    flags the groups of 16 keys concerned and writes their fp32 values beside the halves, and the core multiplies the
    tiles that hold them with fp32 instructions.  Against the restatement, per head of the core's output (the block's
    output projection mixes the heads: it is made a multiple of the identity).  (The scalings stay within what ONE
    power-of-two scale per projection matrix represents: rows 1e5 below the largest keep about 19 bits.  Huge keys make
    scores of 1e5 and a softmax no fp32 evaluation agrees on with another: that case is compared with the other form of
    the library instead, which multiplies the same fp32 values in the same order when K and V of a head are both out of
    range.)"""
    from gpu_util import Ops
    C = 64
    q_rows, k_rows, v_rows = (slice(i * C, (i + 1) * C) for i in range(3))
    head = lambda rows, h: slice(rows.start + 16 * h, rows.start + 16 * h + 16)  # noqa: E731

    def scaled_model():
        model = build_model("cond", 64)
        with torch.no_grad():
            mha = model.get_submodule(prefix).attention
            mha.out_proj.weight.copy_(torch.eye(C))
            mha.out_proj.bias.zero_()
            if case == "huge_values":
                mha.in_proj_weight[head(v_rows, 1)] *= 3e5
                mha.in_proj_bias[head(v_rows, 1)] *= 3e5
            elif case == "tiny_values":    # every head; brought back to order one by the output projection (whose rows are
                mha.in_proj_weight[v_rows] *= 1e-4          # then tiny throughout and take its fp32 form): the residual
                mha.in_proj_bias[v_rows] *= 1e-4            # would hide them
                mha.out_proj.weight.mul_(1e4)
            elif case == "huge_keys_and_values":
                mha.in_proj_weight[head(k_rows, 3)] *= 1e5
                mha.in_proj_bias[head(k_rows, 3)] *= 1e5
                mha.in_proj_weight[head(v_rows, 3)] *= 3e5
                mha.in_proj_bias[head(v_rows, 3)] *= 3e5
        return model

    model = scaled_model()
    sd = state_dict_cpu(model)
    ops = Ops(model.to(dev).hip_engine(dev))
    x = gs.randn(700 + index + batch, batch, C, side, side)
    if case == "nonfinite_value":
        x[batch - 1, 7, 3, 5] = float("inf")   # LayerNorm of this token is NaN: its q, k and v are not finite
    got = ops.attention(index, x.to(dev)).cpu()
    if case == "huge_keys_and_values":
        monkeypatch.setenv("DANRA_PRESPLIT", "0")
        other = Ops(scaled_model().to(dev).hip_engine(dev))
        assert torch.isfinite(got).all()
        assert torch.equal(got, other.attention(index, x.to(dev)).cpu())
        return
    want = cpu_ref.image_self_attention(sd, prefix, x) - x      # the block adds its input: compare the attention part
    got = got - x
    if case == "nonfinite_value":
        # every query of that image meets the token's key: the image is not finite, the others are untouched
        assert not torch.isfinite(got[batch - 1]).any()
        assert rel_err(got[:batch - 1], want[:batch - 1]) < TOL_OP
        return
    assert torch.isfinite(got).all()
    for h in range(4):
        rows = slice(16 * h, 16 * h + 16)
        assert rel_err(got[:, rows], want[:, rows]) < 2 * TOL_OP, (case, h)


@pytest.mark.parametrize("T", [2, 3, 4])
def test_shortest_sampling_chains(cond64, dev, T):
    """T = 2 is one evaluation and a noise-free update, T = 3 the first chain in which the update kernel hands a gathered
    set of time projections to a following evaluation; against the restatement with the same draws."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, sd, _ = cond64
    B = 3
    cpu_in = gs.variant_inputs("cond", B, 64)
    noise = torch.randn(max(T - 2, 0), B, 1, 64, 64, generator=torch.Generator().manual_seed(T))
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    inp = {k: v.to(dev) for k, v in cpu_in.items()}
    got = du.sample_with_host_noise(inp["x"], model, noise.to(dev), inp["y"], cond_img=inp["cond_img"],
                                    lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    draws = iter(noise)
    want = cpu_ref.sample(sd, cpu_in["x"], T, gs.BETA_MIN, gs.BETA_MAX, cpu_in["y"], cpu_in["cond_img"], cpu_in["lsm_cond"],
                          cpu_in["topo_cond"], noise_fn=lambda i, x: next(draws))
    assert rel_err(got.cpu(), want) < TOL_FWD
    again = du.sample_with_host_noise(inp["x"], model, noise.to(dev), inp["y"], cond_img=inp["cond_img"],
                                      lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    assert torch.equal(got, again)  # the loop state (timestep, ticket of the update kernel) starts afresh in every call


def test_split_operand_forward_equals_fp32_products(cond64, golden, dev):
    """danra_set_fp32_products(1) runs every product of the network with fp32 matrix instructions; both forms meet the
    forward tolerance against the reference's own output and agree with each other far inside it."""
    model, _, ops = cond64
    want = torch.from_numpy(golden("forward64")["cond_t500"])
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 2, 64).items()}
    t = torch.full((2,), 500, dtype=torch.long, device=dev)
    outs = {}
    try:
        for mode in (0, 1):
            ops.lib.danra_set_fp32_products(mode)
            with torch.no_grad():
                outs[mode] = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).cpu()
    finally:
        ops.lib.danra_set_fp32_products(0)
    assert rel_err(outs[0], want) < TOL_FWD and rel_err(outs[1], want) < TOL_FWD
    assert rel_err(outs[0], outs[1]) < 2e-5
    assert not torch.equal(outs[0], outs[1]), "the switch changed nothing: the split-operand form did not run"


def test_fp32_products_switch_reaches_a_step_that_was_already_captured(cond64, dev):
    """sample() replays a captured denoise step; flipping danra_set_fp32_products between two calls of the same shape
    must rebuild it (a replay of the old capture would run the other form's kernels)."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, ops = cond64
    B, T = 2, 5
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", B, 64).items()}
    noise = torch.randn(T - 2, B, 1, 64, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")

    def run():
        return du.sample_with_host_noise(inp["x"], model, noise, inp["y"], cond_img=inp["cond_img"],
                                         lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"]).cpu()
    split = run()
    try:
        ops.lib.danra_set_fp32_products(1)
        strict = run()
    finally:
        ops.lib.danra_set_fp32_products(0)
    again = run()
    assert torch.equal(split, again)
    assert not torch.equal(split, strict), "the captured split-operand step was replayed under the fp32 switch"
    assert rel_err(split, strict) < 1e-4


def test_split_operand_k64_chain_falls_back_on_large_activations(cond64, dev):
    """Batch 32 puts the last decoder block's attention tail and the final transposed convolution into one kernel whose
    products are split-operand fp16 products; conditioning planes scaled by 3e5 drive the skip connection it adds - and
    with it the operand of the transposed convolution - far beyond 2^15, so those waves must take the fp32 instructions.
    Both settings of the library agree to the forward tolerance and stay finite."""
    model, _, ops = cond64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 32, 64, seed=12).items()}
    big = {k: (v * 3.0e5 if k in ("cond_img", "topo_cond") else v) for k, v in inp.items()}
    t = torch.full((32,), 250, dtype=torch.long, device=dev)
    outs = {}
    try:
        for mode in (0, 1):
            ops.lib.danra_set_fp32_products(mode)
            with torch.no_grad():
                outs[mode] = model(big["x"], t, big["y"], big["cond_img"], big["lsm_cond"], big["topo_cond"]).cpu()
    finally:
        ops.lib.danra_set_fp32_products(0)
    assert torch.isfinite(outs[0]).all() and torch.isfinite(outs[1]).all()
    assert rel_err(outs[0], outs[1]) < TOL_FWD
    fmaps = model.hip_engine(dev).fmaps()
    assert float(fmaps[0].abs().max()) > 32768.0, "the inputs did not drive the skip connection out of the fp16 range"


def test_odd_batch_on_the_row_stationary_gemm(cond64, dev):
    """B = 33 at 64x64 puts 33792 tokens through the K=64 row-stationary GEMM (QKV at L=1024, final transposed
    convolution) with a partly empty last workgroup; every image must equal its own B = 1 evaluation."""
    model, _, _ = cond64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 33, 64, seed=17).items()}
    t = torch.arange(33, device=dev) * 30 + 1
    with torch.no_grad():
        full = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        for i in (0, 16, 32):
            one = model(inp["x"][i:i + 1], t[i:i + 1], inp["y"][i:i + 1], inp["cond_img"][i:i + 1],
                        inp["lsm_cond"][i:i + 1], inp["topo_cond"][i:i + 1])
            assert rel_err(full[i:i + 1].cpu(), one.cpu()) < TOL_OP, i
        want = cpu_ref.unet_forward(state_dict_cpu(model), inp["x"][32:33].cpu(), t[32:33].cpu(), inp["y"][32:33].cpu(),
                                    inp["cond_img"][32:33].cpu(), inp["lsm_cond"][32:33].cpu(), inp["topo_cond"][32:33].cpu())
    assert rel_err(full[32:33].cpu(), want) < TOL_FWD


@pytest.mark.parametrize("sched", ["linear", "cosine"])
def test_noise_image_bit_exact_and_philox(golden, dev, sched):
    """Forward process kernel (danra_noise_image, used by the training step): bit-exact against the reference's noiseImage given
    its draw; the in-kernel draw is returned, standard normal, reproducible in the seed and consistent with the
    noisy image it produced."""
    from diffusionmodelscustom_amd import DiffusionUtils, hip_engine
    g = golden("noise_image")
    du = DiffusionUtils(gs.NOISE_IMAGE_T, gs.BETA_MIN, gs.BETA_MAX, dev, sched)
    sc = golden("schedule")
    alpha_hat = torch.from_numpy(sc[f"{sched}_{gs.NOISE_IMAGE_T}_alpha_hat"]).to(dev)  # the reference's own table
    x = gs.randn(700, *gs.NOISE_IMAGE_SHAPE).to(dev)
    t = torch.from_numpy(g["t"]).to(dev)
    noise = torch.from_numpy(g[f"{sched}_noise"]).to(dev)
    noisy, z = hip_engine.noise_image(x, t, alpha_hat, noise=noise)
    assert np.array_equal(noisy.cpu().numpy(), g[f"{sched}_noisy"]) and z.data_ptr() == noise.data_ptr()
    # through the class, Philox draws
    xb = torch.randn(64, 1, 64, 64, device=dev)
    tb = du.sampleTimesteps(64)
    du.noise_seed = 5
    n1, z1 = du.noiseImage(xb, tb)
    n2, z2 = du.noiseImage(xb, tb)
    du.noise_seed = 6
    n3, z3 = du.noiseImage(xb, tb)
    assert torch.equal(n1, n2) and torch.equal(z1, z2) and not torch.equal(z1, z3)
    assert abs(z1.mean().item()) < 0.01 and abs(z1.std().item() - 1.0) < 0.01
    keep = torch.sqrt(du.alpha_hat[tb])[:, None, None, None]
    spread = torch.sqrt(1 - du.alpha_hat[tb])[:, None, None, None]
    assert torch.equal(n1, (keep * xb) + (spread * z1))


def test_sampler_state_is_rebuilt_between_calls(cond64, dev):
    """Per-call sampler state (coefficient table, time-projection table, captured step) follows a change of T and of
    the label argument: every call equals the kernel-by-kernel loop and a repeat of itself."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    engine = model.hip_engine(dev)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 2, 64, seed=31).items()}
    results = {}
    try:
        for graph in (True, False):
            engine.set_graph_mode(graph)
            outs = []
            for T, use_y in ((6, True), (9, True), (9, False), (6, True), (1500, True)):
                du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
                steps = 4 if T > 100 else T  # a long table, only its last few timesteps walked
                if T > 100:
                    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "cosine")
                    betas, alphas, ah = du.betas[:steps].clone(), du.alphas[:steps].clone(), du.alpha_hat[:steps].clone()
                else:
                    betas, alphas, ah = du.betas, du.alphas, du.alpha_hat
                outs.append(engine.sample(inp["x"].clone(), inp["y"] if use_y else None, inp["cond_img"], inp["lsm_cond"],
                                          inp["topo_cond"], betas, alphas, ah, seed=3))
            results[graph] = outs
    finally:
        engine.set_graph_mode(True)
    for a, b in zip(results[True], results[False]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert torch.equal(results[True][0], results[True][3])          # same arguments, same result
    assert not torch.equal(results[True][1], results[True][2])      # the label embedding matters


def test_c1_unconditional_32x32_T200_chain(dev):
    """The smallest benchmark configuration (the reference's own CPU-runnable case restated: unconditional 1x32x32, T=200, B=4,
    linear betas): the whole sampling call on the GPU against the CPU restatement with the same host noise, and
    the cosine schedule on the same model."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model = build_model("uncond", 32).to(dev).eval()
    sd = state_dict_cpu(model)
    x0 = gs.randn(55, 4, 1, 32, 32)
    for sched, T in (("linear", 200), ("cosine", 60)):
        du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, sched)
        fn = gs.chain_noise_fn(gs.SEED + 3)
        noise = torch.stack([fn(i, x0) for i in range(T - 1, 1, -1)])
        fn = gs.chain_noise_fn(gs.SEED + 3)
        want = cpu_ref.sample(sd, x0.clone(), T, gs.BETA_MIN, gs.BETA_MAX, scheduler=sched, noise_fn=fn)
        got = du.sample_with_host_noise(x0.to(dev), model, noise.to(dev))
        assert rel_err(got.cpu(), want) < TOL_CHAIN, sched


def test_attention_fallback_shapes(cond64, dev):
    """Shapes beyond the specialised kernels (a 256x256 image reaches L = 1024 at head_dim 32, L = 320 is not a multiple
    of 128 at head_dim 16, L = 272 at head_dim 128 has a ragged last key chunk) take the generic online-softmax kernel."""
    _, sd, ops = cond64
    for seed, idx, name, shape in ((11, 2, "encoder.attention_layers.2", (1, 128, 32, 32)),
                                   (12, 1, "encoder.attention_layers.1", (2, 64, 16, 20)),
                                   (13, 4, "encoder.attention_layers.4", (1, 512, 16, 17))):
        x = gs.randn(seed, *shape)
        want = cpu_ref.image_self_attention(sd, name, x)
        assert rel_err(ops.attention(idx, x.to(dev)).cpu(), want) < TOL_OP, name


@pytest.mark.parametrize("size,batch", [((256, 256), 1), ((96, 160), 3)])
def test_large_and_odd_images_against_the_torch_composition(dev, size, batch):
    """Sizes beyond the golden fixtures: the HIP path against this package's differentiable torch-op composition
    (itself pinned to the reference by the CPU tests) evaluated on the same GPU."""
    H, W = size
    model = build_model("cond", 64).to(dev).eval()   # the lsm/topo buffers' size is not used by forward
    g = torch.Generator().manual_seed(77)
    x = torch.randn(batch, 1, H, W, generator=g).to(dev)
    planes = [torch.randn(batch, 1, H, W, generator=g).to(dev) for _ in range(3)]
    y = torch.randint(0, 4, (batch,), generator=g).to(dev)
    t = torch.randint(1, 1000, (batch,), generator=g).to(dev)
    with torch.no_grad():
        got = model(x, t, y, planes[0], planes[1], planes[2])
    with torch.enable_grad():
        want = model(x, t, y, planes[0], planes[1], planes[2]).detach()
    assert rel_err(got.cpu(), want.cpu()) < TOL_FWD


def test_batch_larger_than_the_benchmark(cond64, dev):
    """B = 130 at 64x64 (twice the benchmark batch, not a multiple of anything): images equal their B = 1 evaluation."""
    model, _, _ = cond64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 130, 64, seed=23).items()}
    t = (torch.arange(130, device=dev) * 7) % 999 + 1
    with torch.no_grad():
        full = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert torch.isfinite(full).all()
        for i in (0, 64, 129):
            one = model(inp["x"][i:i + 1], t[i:i + 1], inp["y"][i:i + 1], inp["cond_img"][i:i + 1],
                        inp["lsm_cond"][i:i + 1], inp["topo_cond"][i:i + 1])
            assert rel_err(full[i:i + 1].cpu(), one.cpu()) < TOL_OP, i


def test_forward_and_chain_on_the_reference_shipped_inputs(cond64, golden, dev):
    """Real conditioning data: ERA5 conditioning images (with their -273.15 fill values), fractional land-sea masks and season
    labels cut from the sample files the reference ships (tests/golden/real_inputs.npz), synthetic topography.
    One forward and a 12-step chain against the restatement."""
    model, sd, _ = cond64
    g = golden("real_inputs")
    cond, lsm = torch.from_numpy(g["cond"]), torch.from_numpy(g["lsm"])
    y = torch.from_numpy(g["season"])
    assert float(cond.min()) < -200.0 and 0.0 <= float(lsm.min()) and float(lsm.max()) <= 1.0 and int(y.max()) <= 3
    topo = gs.variant_inputs("cond", 8, 64)["topo_cond"]
    x = gs.randn(77, 8, 1, 64, 64)
    t = torch.tensor([999, 700, 500, 300, 100, 10, 2, 1])
    with torch.no_grad():
        want = cpu_ref.unet_forward(sd, x, t, y, cond, lsm, topo)
        got = model(x.to(dev), t.to(dev), y.to(dev), cond.to(dev), lsm.to(dev), topo.to(dev))
    assert rel_err(got.cpu(), want) < TOL_FWD
    from diffusionmodelscustom_amd import DiffusionUtils
    T = 13
    fn = gs.chain_noise_fn(5)
    noise = torch.stack([fn(i, x) for i in range(T - 1, 1, -1)])
    fn = gs.chain_noise_fn(5)
    want = cpu_ref.sample(sd, x, T, gs.BETA_MIN, gs.BETA_MAX, y, cond, lsm, topo, noise_fn=fn)
    du = DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    got = du.sample_with_host_noise(x.to(dev), model, noise, y.to(dev), cond_img=cond.to(dev), lsm_cond=lsm.to(dev),
                                    topo_cond=topo.to(dev))
    assert rel_err(got.cpu(), want) < TOL_CHAIN


@pytest.mark.parametrize("layers,dim,out_ch", [([1, 2, 1, 3], 128, 1), ([3, 1, 2, 1], 512, 3), ([2, 2, 2, 2], 64, 2)])
def test_other_geometries(dev, layers, dim, out_ch):
    """Constructor arguments no reference driver uses but the constructors accept: other BasicBlock counts per layer,
    other time-embedding sizes, more than one output channel.  Forward against the restatement."""
    from diffusionmodelscustom_amd import Decoder, DiffusionNet, Encoder
    from diffusionmodelscustom_amd.synthetic import fill_state_dict_
    enc = Encoder(1, dim, lsm_tensor=torch.zeros(1, 64, 64), topo_tensor=torch.zeros(1, 64, 64), cond_on_img=True,
                  cond_img_dim=(1, 64, 64), block_layers=layers, num_classes=4)
    model = fill_state_dict_(DiffusionNet(enc, Decoder(512, out_ch, dim, 64)), 7).eval()
    sd = state_dict_cpu(model)
    model = model.to(dev)
    assert model.hip_supported()
    inp = gs.variant_inputs("cond", 2, 64, seed=31)
    t = torch.tensor([17, 903])
    with torch.no_grad():
        want = cpu_ref.unet_forward(sd, inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        got = model(inp["x"].to(dev), t.to(dev), inp["y"].to(dev), inp["cond_img"].to(dev), inp["lsm_cond"].to(dev),
                    inp["topo_cond"].to(dev))
    assert got.shape == (2, out_ch, 64, 64)
    assert rel_err(got.cpu(), want) < TOL_FWD


def test_fused_final_block_with_three_output_channels(dev):
    """From 32768 token rows on (batch 32 at 64x64) the last decoder block keeps its output at 32x32 and the final 3x3
    convolution rebuilds the transposed convolution in front of it from those rows.  Three output channels make the
    kernel's loop over output channels run more than once; images equal their B = 1 evaluation (which takes the
    unfused kernels) and one of them the restatement."""
    from diffusionmodelscustom_amd import Decoder, DiffusionNet, Encoder
    from diffusionmodelscustom_amd.synthetic import fill_state_dict_
    enc = Encoder(1, 256, lsm_tensor=torch.zeros(1, 64, 64), topo_tensor=torch.zeros(1, 64, 64), cond_on_img=True,
                  cond_img_dim=(1, 64, 64), block_layers=[2, 2, 2, 2], num_classes=4)
    model = fill_state_dict_(DiffusionNet(enc, Decoder(512, 3, 256, 64)), 11).eval()
    sd = state_dict_cpu(model)
    model = model.to(dev)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 32, 64, seed=23).items()}
    t = torch.arange(32, device=dev) * 31 + 2
    with torch.no_grad():
        full = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert full.shape == (32, 3, 64, 64)
        for i in (0, 13, 31):
            one = model(inp["x"][i:i + 1], t[i:i + 1], inp["y"][i:i + 1], inp["cond_img"][i:i + 1],
                        inp["lsm_cond"][i:i + 1], inp["topo_cond"][i:i + 1])
            assert rel_err(full[i:i + 1].cpu(), one.cpu()) < TOL_OP, i
        want = cpu_ref.unet_forward(sd, inp["x"][31:32].cpu(), t[31:32].cpu(), inp["y"][31:32].cpu(),
                                    inp["cond_img"][31:32].cpu(), inp["lsm_cond"][31:32].cpu(), inp["topo_cond"][31:32].cpu())
    assert rel_err(full[31:32].cpu(), want) < TOL_FWD


def test_oversized_call_is_refused_before_any_allocation(cond64, dev):
    """B*H*W >= 2^24 pixels would overflow the kernels' 32-bit offsets: the call is refused with a message."""
    model, _, _ = cond64
    x = torch.zeros(1024, 1, 128, 128, device=dev)
    t = torch.ones(1024, dtype=torch.long, device=dev)
    y = torch.zeros(1024, dtype=torch.long, device=dev)
    with torch.no_grad(), pytest.raises(RuntimeError, match="below 2\\^24 pixels"):
        model(x, t, y, x, x, x)


def test_side_stream_and_two_models_interleaved(cond64, dev):
    """Everything is enqueued on the caller's current stream: a forward and a sampling run issued on a side stream equal
    the default-stream results bit for bit; a second model (its own handle and workspace) used in between does not
    disturb the first one's captured step."""
    from diffusionmodelscustom_amd import DiffusionUtils
    model, _, _ = cond64
    other = build_model("uncond", 64).to(dev)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 3, 64, seed=5).items()}
    t = torch.tensor([5, 500, 995], device=dev)
    du = DiffusionUtils(8, gs.BETA_MIN, gs.BETA_MAX, dev, "linear")
    du.noise_seed = 77
    args = (inp["y"],)
    kw = dict(cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
    with torch.no_grad():
        f0 = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        s0 = du.sample(inp["x"], model, *args, **kw)
        o0 = du.sample(inp["x"], other)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f1 = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
            s1 = du.sample(inp["x"], model, *args, **kw)
            o1 = du.sample(inp["x"], other)
            s2 = du.sample(inp["x"], model, *args, **kw)
        side.synchronize()
    assert torch.equal(f0, f1) and torch.equal(s0, s1) and torch.equal(s0, s2) and torch.equal(o0, o1)
    assert not torch.equal(s0, o0)


def test_unsupported_geometry_on_the_gpu_is_announced(dev):
    """A network the native library does not implement (3 heads) still works on the GPU through the torch composition,
    but says so: the HIP path never degrades silently."""
    from diffusionmodelscustom_amd import Decoder, DiffusionNet, Encoder
    from diffusionmodelscustom_amd.synthetic import fill_state_dict_
    model = fill_state_dict_(DiffusionNet(Encoder(1, 256, n_heads=16), Decoder(512, 1, 256, 64, n_heads=16)), 1).eval().to(dev)
    assert not model.hip_supported()
    x = gs.randn(3, 1, 1, 32, 32).to(dev)
    with torch.no_grad(), pytest.warns(RuntimeWarning, match="not in the HIP library"):
        out = model(x, torch.tensor([5], device=dev))
    assert out.shape == x.shape and torch.isfinite(out).all()


def test_out_of_range_label_is_refused(cond64, dev):
    """nn.Embedding in the reference raises on a label outside the table; so does the binding (the kernels themselves
    clamp, so that a bad label can never become an out-of-bounds read)."""
    model, _, ops = cond64
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 2, 64).items()}
    bad = torch.tensor([1, 4], device=dev)
    with torch.no_grad(), pytest.raises(IndexError, match="out of range"):
        model(inp["x"], torch.tensor([3, 3], device=dev), bad, inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
    # through the raw C entry point the label is clamped to the last class
    t = torch.tensor([9, 9], device=dev)
    _, _, a = ops.time_embed(t, torch.tensor([3, 3], device=dev))
    for wild in (77, 5, -2):
        want = a if wild > 0 else ops.time_embed(t, torch.tensor([3, 0], device=dev))[2]
        _, _, b = ops.time_embed(t, torch.tensor([3, wild], device=dev))
        assert torch.equal(want, b), wild


def test_per_op_entry_points_refuse_bad_arguments(cond64, dev):
    """Null pointers, non-positive sizes and out-of-range selectors come back as DANRA_EINVAL with a message."""
    import ctypes
    model, _, ops = cond64
    lib, h = ops.lib, ops.h
    x = torch.zeros(1, 32, 32, 64, device=dev)
    p = ctypes.c_void_p(x.data_ptr())
    assert lib.danra_attention(h, 9, p, p, 1, 1024, 0, None) == -1
    assert lib.danra_attention(h, 0, None, p, 1, 1024, 0, None) == -1 and b"null" in lib.danra_last_error()
    assert lib.danra_attention(h, 0, p, p, 0, 1024, 0, None) == -1
    assert lib.danra_resnet_layer(h, 5, p, p, 1, 16, 16, None) == -1
    assert lib.danra_resnet_layer(h, 1, p, p, 1, -16, 16, None) == -1
    assert lib.danra_decoder_block(h, 2, p, None, None, p, 1, 8, 8, None) == -1 and b"skip" in lib.danra_last_error()
    assert lib.danra_instance_norm(p, p, 1, 1024, 48, None) == -1
    assert lib.danra_conv1(h, p, p, p, p, p, 1, 40, 64, None) == -1 and b"multiples of 16" in lib.danra_last_error()
    assert lib.danra_nchw_to_nhwc(p, None, 1, 64, 1024, None) == -1


def test_128x128_batch_32_equals_single_image_evaluations(dev):
    """The 128x128 benchmark shape: at batch 32 both 64-channel decoder blocks hand their output to the next transposed
    convolution inside one kernel (with its InstanceNorm slice statistics); at batch 1 neither does."""
    model = build_model("cond", 128).to(dev)
    inp = {k: v.to(dev) for k, v in gs.variant_inputs("cond", 32, 128, seed=3).items()}
    t = (torch.arange(32, device=dev) * 31) % 999 + 1
    with torch.no_grad():
        full = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        assert torch.isfinite(full).all()
        for i in (0, 17, 31):
            one = model(inp["x"][i:i + 1], t[i:i + 1], inp["y"][i:i + 1], inp["cond_img"][i:i + 1],
                        inp["lsm_cond"][i:i + 1], inp["topo_cond"][i:i + 1])
            assert rel_err(full[i:i + 1].cpu(), one.cpu()) < TOL_OP, i
