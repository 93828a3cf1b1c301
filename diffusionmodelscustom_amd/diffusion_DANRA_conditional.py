"""DDPM process utilities with the call surface of the reference's ``diffusion_DANRA_conditional.py``
(DDPM_DANRA_conditional/diffusion_DANRA_conditional.py:13-159).

``sample()`` hands the whole reverse loop to ``libdanra_hip.so`` (one C call, no host work per
step) whenever the model is this package's ``DiffusionNet`` on a CUDA device; with any other model or
on CPU tensors it runs the generic step loop through ``model(...)``, as the reference does.
"""
import numpy as np
import torch
import torch.nn as nn

from .modules_DANRA_conditional import *  # noqa: F401,F403  (the reference re-exports its modules the same way)
from .modules_DANRA_conditional import DiffusionNet

try:  # progress bar as in the reference; optional
    import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


class DiffusionUtils:
    """Gaussian diffusion helper: beta schedule, forward noising, reverse sampling."""

    def __init__(self, n_timesteps: int, beta_min: float, beta_max: float, device: str = 'cpu',
                 scheduler: str = 'linear'):
        assert scheduler in ['linear', 'cosine'], 'scheduler must be linear or cosine'
        self.n_timesteps = n_timesteps
        self.beta_min = beta_min
        self.beta_max = beta_max
        self.device = device
        self.scheduler = scheduler
        self.betas = self.betaSamples().to(self.device)
        self.alphas = 1 - self.betas
        self.alpha_hat = torch.cumprod(self.alphas, dim=0)
        # sampling options of this implementation (the reference has none)
        self.show_progress = False
        self.noise_seed = None      # None: derive the Philox seed from torch's generator at each sample() call
        self._host_noise = None     # set by sample_with_host_noise()
        self.first_image = 0        # index of x[0] in the global batch when this process samples one slice of it

    def betaSamples(self):
        """Linear schedule, or the reference's half-cosine ramp from beta_min to beta_max
        (reference diffusion:53-77; evaluated in double, stored fp32)."""
        if self.scheduler == 'linear':
            return torch.linspace(start=self.beta_min, end=self.beta_max, steps=self.n_timesteps).to(self.device)
        last = self.n_timesteps - 1
        ramp = [self.beta_min + 0.5 * (self.beta_max - self.beta_min) * (1 + np.cos((i / last) * np.pi))
                for i in reversed(range(self.n_timesteps))]
        return torch.Tensor(ramp).to(self.device)

    def sampleTimesteps(self, size: int):
        return torch.randint(low=1, high=self.n_timesteps, size=(size,)).to(self.device)

    def noiseImage(self, x: torch.Tensor, t: torch.LongTensor, noise: torch.Tensor = None):
        """Forward process (reference diffusion:85-103): returns (noisy image, the noise that was added).
        On a CUDA fp32 batch whose per-image size is a multiple of 4 this is one HIP kernel, the Gaussian draws made
        in-kernel (Philox) unless `noise` supplies them; elsewhere the reference's torch expressions."""
        assert len(x.shape) == 4, 'x must be a 4D tensor'
        if (x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and x[0].numel() % 4 == 0
                and self.alpha_hat.device == x.device):
            from . import hip_engine
            return hip_engine.noise_image(x, t, self.alpha_hat, noise=noise, seed=self._philox_seed(), offset=0)
        keep = torch.sqrt(self.alpha_hat[t])[:, None, None, None]
        spread = torch.sqrt(1 - self.alpha_hat[t])[:, None, None, None]
        if noise is None:
            noise = torch.randn_like(x).to(self.device)
        return (keep * x) + (spread * noise), noise

    # ------------------------------------------------------------------ reverse process
    def sample(self, x: torch.Tensor, model: nn.Module, y: torch.Tensor = None, cond_img: torch.Tensor = None,
               lsm_cond: torch.Tensor = None, topo_cond: torch.Tensor = None):
        assert len(x.shape) == 4, 'x must be a 4D tensor'
        model.eval()
        with torch.no_grad():
            if isinstance(model, DiffusionNet) and model._hip_eligible(x):
                return self._sample_hip(x, model, y, cond_img, lsm_cond, topo_cond, self._host_noise)
            return self._sample_generic(x, model, y, cond_img, lsm_cond, topo_cond, self._host_noise)

    def sample_with_host_noise(self, x, model, noise, y=None, cond_img=None, lsm_cond=None, topo_cond=None):
        """``sample()`` with the per-step Gaussian draws supplied by the caller: ``noise[k]`` is used at
        timestep index ``n_timesteps-1-k`` (k = 0 .. n_timesteps-3; the last step adds no noise).  This is how
        a run is made comparable with a CPU run of the reference, whose ``randn_like`` draws come from the
        host generator."""
        self._host_noise = noise
        try:
            return self.sample(x, model, y, cond_img=cond_img, lsm_cond=lsm_cond, topo_cond=topo_cond)
        finally:
            self._host_noise = None

    def _philox_seed(self):
        if self.noise_seed is not None:
            return int(self.noise_seed)
        return int(torch.randint(0, 2 ** 62, (1,)).item())  # follows torch.manual_seed, like randn_like would

    def _sample_hip(self, x, model, y, cond_img, lsm_cond, topo_cond, host_noise):
        engine = model.hip_engine(x.device)
        x = x.detach().clone().contiguous()  # the reference returns a new tensor and leaves its input untouched
        if host_noise is not None:
            host_noise = host_noise.to(device=x.device, dtype=torch.float32)
            assert host_noise.shape[0] >= self.n_timesteps - 2, 'need one noise tensor per step with i > 1'
        return engine.sample(x, *model.hip_call_arguments(y, cond_img, lsm_cond, topo_cond),
                             self.betas, self.alphas, self.alpha_hat, noise=host_noise, seed=self._philox_seed(),
                             first_image=self.first_image)

    def _sample_generic(self, x, model, y, cond_img, lsm_cond, topo_cond, host_noise):
        steps = reversed(range(1, self.n_timesteps))
        if self.show_progress and _tqdm is not None:
            steps = _tqdm.tqdm(steps)
        draw = 0
        for i in steps:
            t = (torch.ones(x.shape[0]) * i).long().to(self.device)
            alpha = self.alphas[t][:, None, None, None]
            beta = self.betas[t][:, None, None, None]
            alpha_hat = self.alpha_hat[t][:, None, None, None]
            predicted_noise = model(x, t, y, cond_img, lsm_cond, topo_cond)
            if i > 1:
                if host_noise is not None:
                    noise = host_noise[draw].to(x.device)
                    draw += 1
                else:
                    noise = torch.randn_like(x).to(self.device)
            else:
                noise = torch.zeros_like(x).to(self.device)
            x = (1 / torch.sqrt(alpha)) * (x - ((1 - alpha) / torch.sqrt(1 - alpha_hat)) * predicted_noise)
            x = x + (torch.sqrt(beta) * noise)
        return x
