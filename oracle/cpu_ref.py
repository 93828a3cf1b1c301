"""CPU restatement of the reference's DDPM sampling + UNet forward path.

TEST INFRASTRUCTURE ONLY: the checker that the HIP path is compared against, and the timed
CPU baseline of ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does.

It is a functional (state_dict in, tensors out) restatement in torch CPU fp32 ops of
/root/reference/DDPM_DANRA_conditional/{diffusion,modules}_DANRA_conditional.py, each function
citing the lines it follows.  Functions taking ``clean=True`` (and those named ``*_clean``) restate the
variant in /root/reference/DDPM_clean_application/src/{unet,diffusion_modules}.py instead.  Parity pin: ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/*.npz``, which ``oracle/make_golden.py`` produced by importing and
running the reference's own code in the build container (the reference holds no tests or golden
vectors of its own for this path).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FMAP_CHANNELS = (64, 64, 128, 256, 512)  # modules_DANRA_conditional.py:170


# --------------------------------------------------------------------------- schedule
def beta_schedule(n_timesteps, beta_min, beta_max, scheduler="linear"):
    """diffusion_DANRA_conditional.py:53-77 (betaSamples)."""
    assert scheduler in ("linear", "cosine"), "scheduler must be linear or cosine"
    if scheduler == "linear":
        return torch.linspace(beta_min, beta_max, n_timesteps)
    betas = []
    T = n_timesteps - 1
    for i in reversed(range(n_timesteps)):
        betas.append(beta_min + 0.5 * (beta_max - beta_min) * (1 + np.cos((i / T) * np.pi)))
    return torch.Tensor(betas)


def schedule(n_timesteps, beta_min, beta_max, scheduler="linear"):
    """betas, alphas, alpha_hat as in diffusion_DANRA_conditional.py:47-51."""
    betas = beta_schedule(n_timesteps, beta_min, beta_max, scheduler)
    alphas = 1 - betas
    alpha_hat = torch.cumprod(alphas, dim=0)
    return betas, alphas, alpha_hat


def update_coefficients(betas, alphas, alpha_hat, i):
    """The three fp32 scalars of diffusion_DANRA_conditional.py:135-157 for timestep index i:
    c1 = 1/sqrt(alpha), c2 = (1-alpha)/sqrt(1-alpha_hat), c3 = sqrt(beta)."""
    a, b, ah = alphas[i], betas[i], alpha_hat[i]
    return 1 / torch.sqrt(a), (1 - a) / torch.sqrt(1 - ah), torch.sqrt(b)


def ddpm_update(x, eps, noise, c1, c2, c3):
    """diffusion_DANRA_conditional.py:155,157 (two statements, every op rounded to fp32)."""
    x = c1 * (x - c2 * eps)
    return x + c3 * noise


def noise_image(x, t, alpha_hat, noise):
    """diffusion_DANRA_conditional.py:96-103 given the Gaussian draw: sqrt(alpha_hat[t]) * x + sqrt(1 - alpha_hat[t]) * noise,
    per-image coefficients broadcast over (C, H, W)."""
    keep = torch.sqrt(alpha_hat[t])[:, None, None, None]
    spread = torch.sqrt(1 - alpha_hat[t])[:, None, None, None]
    return (keep * x) + (spread * noise)


# --------------------------------------------------------------------------- embeddings
def encoder_pos_encoding(t, channels=256):
    """modules_DANRA_conditional.py:203-211,243-244: base 1000, [sin x128 | cos x128]."""
    t = t.unsqueeze(-1).type(torch.float)
    inv_freq = 1.0 / (1000 ** (torch.arange(0, channels, 2).float() / channels))
    arg = t.repeat(1, channels // 2) * inv_freq
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def decoder_sinusoidal_embedding(t, dim=256, n=10000, faithful=False):
    """modules_DANRA_conditional.py:42-63: base 10000, interleaved sin/cos; the integer timestep is
    divided by a Python-double power that torch rounds to fp32 before the fp32 division."""
    if faithful:  # the reference's per-element Python loop, verbatim semantics
        out = torch.zeros(len(t), dim)
        for idx in range(len(t)):
            for i in range(dim // 2):
                emb = t[idx] / (n ** (2 * i / dim))
                out[idx, 2 * i] = torch.sin(emb)
                out[idx, 2 * i + 1] = torch.cos(emb)
        return out
    div = torch.tensor([n ** (2 * i / dim) for i in range(dim // 2)], dtype=torch.float64).float()
    emb = t.to(torch.float32)[:, None] / div[None, :]
    return torch.stack([torch.sin(emb), torch.cos(emb)], dim=-1).reshape(len(t), dim)


def time_projection(sd, prefix, emb):
    """Sequential(SiLU, Linear): modules_DANRA_conditional.py:325-329,401-404."""
    return F.linear(F.silu(emb), sd[prefix + ".1.weight"], sd[prefix + ".1.bias"])


# --------------------------------------------------------------------------- blocks
def image_self_attention(sd, prefix, x, n_heads=4, faithful=False):
    """modules_DANRA_conditional.py:91-110: LayerNorm -> 4-head self-attention -> + un-normalised x."""
    B, C, H, W = x.shape
    tok = x.reshape(-1, C, H * W).permute(0, 2, 1)
    nx = F.layer_norm(tok, (C,), sd[prefix + ".layernorm.weight"], sd[prefix + ".layernorm.bias"], 1e-5)
    qkv = F.linear(nx, sd[prefix + ".attention.in_proj_weight"], sd[prefix + ".attention.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    hd = C // n_heads
    L = H * W

    def heads(z):
        return z.reshape(B, L, n_heads, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    scores = torch.matmul(q * (1.0 / math.sqrt(hd)), k.transpose(-1, -2))
    p = torch.softmax(scores, dim=-1)
    if faithful:  # need_weights=True also averages the probabilities over heads and discards them
        _ = p.mean(dim=1)
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, L, C)
    o = F.linear(o, sd[prefix + ".attention.out_proj.weight"], sd[prefix + ".attention.out_proj.bias"])
    o = o + tok
    return o.permute(0, 2, 1).reshape(-1, C, H, W)


def image_self_attention_clean(sd, prefix, x, n_heads=4):
    """DDPM_clean_application/src/unet.py:98-119: LayerNorm -> self-attention (`.mha`) -> + x, then
    + ff(.) with ff = LayerNorm -> Linear -> exact GELU -> Linear (:91-96)."""
    B, C, H, W = x.shape
    mapped = {prefix + ".layernorm.weight": sd[prefix + ".layernorm.weight"],
              prefix + ".layernorm.bias": sd[prefix + ".layernorm.bias"]}
    for k in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"):
        mapped[prefix + ".attention." + k] = sd[prefix + ".mha." + k]
    a = image_self_attention(mapped, prefix, x, n_heads)
    tok = a.reshape(-1, C, H * W).permute(0, 2, 1)
    h = F.layer_norm(tok, (C,), sd[prefix + ".ff.0.weight"], sd[prefix + ".ff.0.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[prefix + ".ff.1.weight"], sd[prefix + ".ff.1.bias"]))
    h = F.linear(h, sd[prefix + ".ff.3.weight"], sd[prefix + ".ff.3.bias"])
    return (tok + h).permute(0, 2, 1).reshape(-1, C, H, W)


def _attention(sd, prefix, x, clean, faithful=False, n_heads=4):
    if clean:
        return image_self_attention_clean(sd, prefix, x, n_heads)
    return image_self_attention(sd, prefix, x, n_heads, faithful=faithful)


def _bn(sd, prefix, x):
    """BatchNorm2d in eval mode (diffusion_DANRA_conditional.py:122 puts the model in eval)."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, 1e-5)


def basic_block(sd, prefix, x, stride):
    """torchvision 0.16.1 BasicBlock.forward (see oracle/torchvision_standin.py)."""
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, prefix + ".bn1", out))
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, prefix + ".bn2", out)
    if prefix + ".downsample.0.weight" in sd:
        identity = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0)
        identity = _bn(sd, prefix + ".downsample.1", identity)
    else:
        identity = x
    return F.relu(out + identity)


def resnet_layer(sd, prefix, x, stride):
    n = 0
    while f"{prefix}.{n}.conv1.weight" in sd:
        x = basic_block(sd, f"{prefix}.{n}", x, stride if n == 0 else 1)
        n += 1
    return x


def encoder_forward(sd, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None, faithful=False, clean=False,
                    n_heads=4):
    """modules_DANRA_conditional.py:213-312; clean: DDPM_clean_application/src/unet.py:217-316, where lsm/topo are
    concatenated iff passed (:232-239) instead of iff the buffers exist."""
    p = "encoder."
    if (lsm_cond is not None) if clean else (p + "lsm" in sd):
        x = torch.cat([x, lsm_cond], dim=1)
    if (topo_cond is not None) if clean else (p + "elevation" in sd):
        x = torch.cat([x, topo_cond], dim=1)
    if cond_img is not None:
        x = torch.cat((x, cond_img), dim=1)
    emb = encoder_pos_encoding(t, sd[p + "time_projection_layers.0.1.weight"].shape[1])  # = time_embedding
    if y is not None:
        emb = emb + F.embedding(y, sd[p + "label_emb.weight"])
    fmaps = []
    f = F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)
    f = f + time_projection(sd, p + "time_projection_layers.0", emb)[:, :, None, None]
    f = _attention(sd, p + "attention_layers.0", f, clean, faithful, n_heads)
    fmaps.append(f)
    h = F.conv2d(f, sd[p + "conv2.weight"], None, 2, 3)
    h = F.relu(_bn(sd, p + "bn1", h))
    for lvl in range(1, 5):
        h = resnet_layer(sd, f"{p}layer{lvl}", h, 1 if lvl == 1 else 2)
        h = h + time_projection(sd, f"{p}time_projection_layers.{lvl}", emb)[:, :, None, None]
        h = _attention(sd, f"{p}attention_layers.{lvl}", h, clean, faithful, n_heads)
        fmaps.append(h)
    return tuple(fmaps)


def decoder_block(sd, prefix, fmap, prev_fmap=None, t=None, final=False, faithful=False, clean=False, n_heads=4):
    """modules_DANRA_conditional.py:425-460; ``final`` = the block built at :503-509
    (no second InstanceNorm, no attention, Identity activation)."""
    out = F.conv_transpose2d(fmap, sd[prefix + ".transpose.weight"], sd[prefix + ".transpose.bias"], stride=2)
    out = F.instance_norm(out, eps=1e-5)
    out = F.conv2d(out, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], 1, 1)
    if not final:
        out = F.instance_norm(out, eps=1e-5)
    if prev_fmap is not None:
        assert prev_fmap.shape == out.shape, "feature maps must be of same shape"
        out = out + prev_fmap
    if t is not None:
        emb = decoder_sinusoidal_embedding(t, dim=sd[prefix + ".time_projection_layer.1.weight"].shape[1],
                                           faithful=faithful)
        out = out + time_projection(sd, prefix + ".time_projection_layer", emb)[:, :, None, None]
        out = _attention(sd, prefix + ".attention", out, clean, faithful, n_heads)
    return out if final else F.relu(out)


def decoder_forward(sd, fmaps, t, faithful=False, clean=False, n_heads=4):
    """modules_DANRA_conditional.py:512-536 (clean: unet.py:516-540, the same)."""
    fm = list(reversed(fmaps))
    out = decoder_block(sd, "decoder.residual_layers.0", fm[0], fm[1], t, faithful=faithful, clean=clean, n_heads=n_heads)
    for idx in range(1, 4):
        out = decoder_block(sd, f"decoder.residual_layers.{idx}", out, fm[idx + 1], t, faithful=faithful, clean=clean,
                            n_heads=n_heads)
    return decoder_block(sd, "decoder.final_layer", out, final=True)


def unet_forward(sd, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None, faithful=False, clean=False,
                 n_heads=4):
    """DiffusionNet.forward, modules_DANRA_conditional.py:597-616 (clean: unet.py:597-616)."""
    fmaps = encoder_forward(sd, x, t, y, cond_img, lsm_cond, topo_cond, faithful=faithful, clean=clean, n_heads=n_heads)
    return decoder_forward(sd, fmaps, t, faithful=faithful, clean=clean, n_heads=n_heads)


# --------------------------------------------------------------------------- sampling loop
def sample(sd, x, n_timesteps, beta_min, beta_max, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
           scheduler="linear", noise_fn=None, trace=None, faithful=False):
    """DiffusionUtils.sample, diffusion_DANRA_conditional.py:105-159.

    ``noise_fn(i, x)`` supplies the per-step Gaussian draw (default ``torch.randn_like``, which on
    CPU consumes the global generator in the reference's draw order); ``trace`` (dict) receives x
    after the update of every timestep index listed in ``trace['want']``."""
    assert len(x.shape) == 4, "x must be a 4D tensor"
    betas, alphas, alpha_hat = schedule(n_timesteps, beta_min, beta_max, scheduler)
    if noise_fn is None:
        noise_fn = lambda i, x_: torch.randn_like(x_)  # noqa: E731
    with torch.no_grad():
        for i in reversed(range(1, n_timesteps)):
            t = (torch.ones(x.shape[0]) * i).long()
            alpha = alphas[t][:, None, None, None]
            beta = betas[t][:, None, None, None]
            ah = alpha_hat[t][:, None, None, None]
            eps = unet_forward(sd, x, t, y, cond_img, lsm_cond, topo_cond, faithful=faithful)
            noise = noise_fn(i, x) if i > 1 else torch.zeros_like(x)
            x = (1 / torch.sqrt(alpha)) * (x - ((1 - alpha) / torch.sqrt(1 - ah)) * eps)
            x = x + torch.sqrt(beta) * noise
            if trace is not None and i in trace.get("want", ()):
                trace[i] = x.clone()
    return x


# --------------------------------------------------------------------------- clean-application sampler
def beta_schedule_clean(n_timesteps, beta_min, beta_max, scheduler="linear"):
    """DDPM_clean_application/src/diffusion_modules.py:51-69: linear, or 1 - abar_t/abar_{t-1} of the squared-cosine
    abar, clipped to [1e-4, 0.9999]."""
    assert scheduler in ("linear", "cosine"), "scheduler must be linear or cosine"
    if scheduler == "linear":
        return torch.linspace(beta_min, beta_max, n_timesteps)
    t = torch.linspace(0, n_timesteps, n_timesteps + 1)
    ft = torch.cos(((t / n_timesteps + 0.008) / 1.008) * np.pi / 2) ** 2
    alphat = ft / ft[0]
    betat = 1 - alphat[1:] / alphat[:-1]
    return torch.clip(betat, 0.0001, 0.9999)


def sample_clean(sd, x, n_timesteps, beta_min, beta_max, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
                 scheduler="linear", cfg_scale=0.0, data_scaled=False, noise_fn=None, trace=None):
    """DiffusionUtils.sample, DDPM_clean_application/src/diffusion_modules.py:101-186, from the start image on
    (``x`` is the randn draw of :129; the 0.005 shrink of :131-132 is applied here)."""
    betas = beta_schedule_clean(n_timesteps, beta_min, beta_max, scheduler)
    alphas = 1 - betas
    alpha_hat = torch.cumprod(alphas, dim=0)
    if noise_fn is None:
        noise_fn = lambda i, x_: torch.randn_like(x_)  # noqa: E731
    with torch.no_grad():
        if data_scaled:
            x = x * 0.005
        for i in reversed(range(1, n_timesteps)):
            t = (torch.ones(x.shape[0]) * i).long()
            eps = unet_forward(sd, x, t, y, cond_img, lsm_cond, topo_cond, clean=True)
            if cfg_scale > 0:
                eps = torch.lerp(unet_forward(sd, x, t, clean=True), eps, cfg_scale)
            alpha = alphas[t][:, None, None, None]
            beta = betas[t][:, None, None, None]
            ah = alpha_hat[t][:, None, None, None]
            if i > 1:
                noise = noise_fn(i, x)
                if data_scaled:
                    noise = noise * 0.005
            else:
                noise = torch.zeros_like(x)
            x = (1 / torch.sqrt(alpha)) * (x - ((1 - alpha) / torch.sqrt(1 - ah)) * eps)
            x = x + torch.sqrt(beta) * noise
            if trace is not None and i in trace.get("want", ()):
                trace[i] = x.clone()
    return x
