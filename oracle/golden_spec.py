"""Case list and seeded inputs shared by ``oracle/make_golden.py`` (which runs the reference)
and the tests (which run the CPU restatement and the HIP path on the same inputs).

TEST INFRASTRUCTURE ONLY.  Inputs are regenerated from seeds instead of being stored, so the
committed fixtures hold only what the reference computed.
"""
import torch

from diffusionmodelscustom_amd.synthetic import synthetic_inputs

SEED = 42
TIME_EMBEDDING = 256

# Model variants exercised by the reference drivers (generation_unconditional.py:133-141,
# generation_lsm_topo_only.py:137-142, generation_DANRA_conditional.py:280-287).
VARIANTS = {
    "uncond": dict(lsm=False, topo=False, cond=False, num_classes=None),
    "lsm_topo": dict(lsm=True, topo=True, cond=False, num_classes=None),
    "cond": dict(lsm=True, topo=True, cond=True, num_classes=4),
}

SCHEDULE_CASES = [(T, s) for T in (200, 800, 1000) for s in ("linear", "cosine")]
EMBED_TIMESTEPS = (1, 2, 5, 137, 999)
NOISE_IMAGE_T = 1000
NOISE_IMAGE_SHAPE = (6, 1, 32, 32)
NOISE_IMAGE_TIMESTEPS = (1, 2, 137, 500, 998, 999)
FORWARD_TIMESTEPS = (1, 500, 999)
ATTN_LEVELS = ((0, 64, 32), (1, 64, 16), (2, 128, 8), (3, 256, 4), (4, 512, 2))  # (level, C, side)
CHAIN_CASES = {"T50": dict(T=50, B=2), "T200": dict(T=200, B=2), "T1000": dict(T=1000, B=1)}
BETA_MIN, BETA_MAX = 1e-4, 0.02

# The DDPM_clean_application network (src/unet.py) as src/main_ddpm.py:354-369 builds it (planes + 4 seasons) and a
# label-only one, the only kind its classifier-free guidance can run on (src/diffusion_modules.py:158 passes no planes).
CLEAN_VARIANTS = {
    "clean_cond": dict(lsm=True, topo=True, cond=True, num_classes=4),
    "clean_class": dict(lsm=False, topo=False, cond=False, num_classes=4),
}
CLEAN_SCHEDULE_T = (50, 200, 1000)
CLEAN_ATTN_LEVELS = ((0, 64, 32), (3, 256, 4))
CLEAN_CHAINS = {
    "plain": dict(variant="clean_cond", T=50, B=2, scheduler="linear", cfg_scale=0.0, data_scaled=False),
    "guided": dict(variant="clean_class", T=50, B=2, scheduler="linear", cfg_scale=3.0, data_scaled=True),
    "guided_small": dict(variant="clean_class", T=50, B=2, scheduler="cosine", cfg_scale=0.3, data_scaled=False),
}
CLEAN_CHAIN_CHECKPOINTS = (49, 25, 1)
# head counts other than the constructors' default of 4: the clean application's configuration default is 1
# (DDPM_clean_application/config.yaml:51, test/launch.py:62); 2 and 8 exercise head_dim 32 and 8 at the 64-channel levels
HEAD_CASES = (("clean_cond", 1), ("clean_class", 2), ("cond", 8))


def randn(seed, *shape):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


def variant_inputs(variant, batch, size, seed=SEED):
    """Forward-call arguments (x, y, cond_img, lsm_cond, topo_cond) for a model variant."""
    v = VARIANTS[variant] if variant in VARIANTS else CLEAN_VARIANTS[variant]
    d = synthetic_inputs(batch, size, size, seed=seed, conditional=True)
    return dict(
        x=d["x"],
        y=d["y"] if v["num_classes"] else None,
        cond_img=d["cond_img"] if v["cond"] else None,
        lsm_cond=d["lsm_cond"] if v["lsm"] else None,
        topo_cond=d["topo_cond"] if v["topo"] else None,
    )


def chain_checkpoints(T):
    """Timestep indices after whose update x is recorded (diffusion_DANRA_conditional.py:130-157)."""
    return sorted({T - 1, T - 2, T // 2, 2, 1}, reverse=True)


def chain_noise_fn(seed):
    """Per-step host noise in the reference's draw order: one ``randn`` of x's shape per step with
    i > 1, from a CPU generator seeded once (diffusion_DANRA_conditional.py:149-152 with
    torch.manual_seed as in generation_DANRA_conditional.py:41)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def fn(i, x):
        return torch.randn(x.shape, generator=g)

    return fn


def clean_chain_draws(batch, size, seed=SEED + 2):
    """Start image and per-step noise of a DDPM_clean_application sampling run in the reference's draw order:
    ``torch.randn((n, 1, size, size))`` (diffusion_modules.py:129), then one ``randn_like`` per step with i > 1 (:172),
    all from one generator seeded once."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.randn((batch, 1, size, size), generator=g)

    def fn(i, x_):
        return torch.randn(x_.shape, generator=g)

    return x, fn
