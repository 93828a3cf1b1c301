"""Deterministic synthetic weights and inputs.

No checkpoints or data files ship with the reference, so every test, golden vector and
benchmark uses (a) a weight filler keyed by state_dict entry name and (b) seeded synthetic
inputs whose distributions follow the saved samples of the reference
(DDPM_DANRA_conditional/final_generated_samples/*.npz: conditioning temperatures in about
[-10, 26] with mean 8.8, land-sea mask mean 0.45) and DDPM_clean_application/test/unet_test.py:76
(topography range).  The same functions fill the reference model (when it is imported to produce
golden vectors) and this package's model, so both hold bit-identical parameters.
"""
import zlib

import torch

__all__ = ["fill_state_dict_", "synthetic_state_dict", "synthetic_inputs", "SyntheticDANRADataset"]


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _fill_one(name: str, ref: torch.Tensor, seed: int) -> torch.Tensor:
    g = _gen(name, seed)
    shape = tuple(ref.shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=ref.dtype)
    if name.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if name.endswith("running_mean"):
        return 0.1 * torch.randn(shape, generator=g)
    if name.endswith(("encoder.lsm", "encoder.elevation")):
        return ref.detach().clone().cpu()  # only their existence matters
    if "label_emb" in name:
        return 0.5 * torch.randn(shape, generator=g)
    if ref.dim() == 1:
        if name.endswith("weight"):            # BatchNorm / LayerNorm scale
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)  # every kind of bias
    fan_in = max(1, ref.numel() // shape[0])
    return torch.randn(shape, generator=g) / float(fan_in) ** 0.5


def synthetic_state_dict(state_dict, seed: int = 42):
    """Return a new CPU fp32 state_dict with the same keys/shapes, deterministically filled."""
    out = {}
    for name, ref in state_dict.items():
        v = _fill_one(name, ref, seed)
        out[name] = v.to(ref.dtype) if ref.dtype != torch.float32 else v.float()
    return out


def fill_state_dict_(module: torch.nn.Module, seed: int = 42) -> torch.nn.Module:
    """Overwrite every parameter and buffer of ``module`` in place with the synthetic values."""
    module.load_state_dict(synthetic_state_dict(module.state_dict(), seed), strict=True)
    return module


def synthetic_inputs(batch: int, height: int, width: int, seed: int = 42, conditional: bool = True,
                     num_classes: int = 4):
    """x_T ~ N(0,1); y ~ randint(num_classes); cond ~ N(8, 6^2); lsm ~ Bernoulli(0.45);
    topo ~ U(-12, 330).  Returned on CPU; unscaled, as the reference drivers pass them
    (scale=False, DDPM_DANRA_conditional_lumi/ddpm_DANRA_conditional_wValid__128x128.py:196)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.randn(batch, 1, height, width, generator=g)
    if not conditional:
        return {"x": x, "y": None, "cond_img": None, "lsm_cond": None, "topo_cond": None}
    y = torch.randint(0, num_classes, (batch,), generator=g)
    cond = 8.0 + 6.0 * torch.randn(batch, 1, height, width, generator=g)
    lsm = (torch.rand(batch, 1, height, width, generator=g) < 0.45).float()
    topo = -12.0 + 342.0 * torch.rand(batch, 1, height, width, generator=g)
    return {"x": x, "y": y, "cond_img": cond, "lsm_cond": lsm, "topo_cond": topo}


class SyntheticDANRADataset(torch.utils.data.Dataset):
    """Items with the dict contract of the reference's cut-out dataset
    (DDPM_DANRA_conditional_lumi/data_DANRA_conditional.py:513-564): ``img`` (the high-resolution truth, [1, H, W]),
    ``classifier`` (season label, int64 scalar), ``img_cond`` (conditioning image), ``lsm``, ``sdf``, ``topo`` (cut-outs of
    the static fields) and ``points`` (the cut-out corners), each key present only when the corresponding option is on,
    as there.  Values are seeded synthetic fields (see ``synthetic_inputs``); nothing is read from disk."""

    def __init__(self, n_samples: int, size: int, conditional_seasons=True, conditional_images=True, lsm_topo=True,
                 sdf_weighted_loss=False, num_classes: int = 4, seed: int = 42):
        self.n_samples, self.size, self.seed, self.num_classes = n_samples, size, seed, num_classes
        self.conditional_seasons, self.conditional_images = conditional_seasons, conditional_images
        self.lsm_topo, self.sdf_weighted_loss = lsm_topo, sdf_weighted_loss

    def __len__(self):
        return self.n_samples

    def __getitem__(self, idx):
        d = synthetic_inputs(1, self.size, self.size, seed=self.seed + 1000 * idx, num_classes=self.num_classes)
        sample = {'img': d["x"][0]}
        if self.conditional_images:
            if self.conditional_seasons:
                sample['classifier'] = d["y"][0]
            sample['img_cond'] = d["cond_img"][0]
        if self.lsm_topo:
            sample['lsm'] = d["lsm_cond"][0]
            if self.sdf_weighted_loss:
                sample['sdf'] = torch.rand(1, self.size, self.size, generator=_gen("sdf", self.seed + idx))
            sample['topo'] = d["topo_cond"][0]
        sample['points'] = torch.tensor([0, self.size, 0, self.size])
        return sample
