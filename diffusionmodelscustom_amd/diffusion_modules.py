"""DDPM process utilities with the call surface of the reference's
``DDPM_clean_application/src/diffusion_modules.py`` (:7-186).

Relative to ``diffusion_DANRA_conditional.DiffusionUtils`` the sampler draws its own start image
(``sample(n, model, channels_hr, ...)``), supports classifier-free guidance (``cfg_scale``), can shrink every
Gaussian draw by 0.005 (``data_scaled``), uses the "improved DDPM" cosine schedule, and puts the model back into
training mode when it returns.  On a CUDA device with this package's ``unet.DiffusionNet`` (or the DANRA one) the
whole reverse loop is one call into ``libdanra_hip.so``; otherwise the generic step loop runs through ``model(...)``.
"""
import numpy as np
import torch
import torch.nn as nn

from .modules_DANRA_conditional import DiffusionNet as _HipCapableNet

try:  # progress bar as in the reference; optional
    import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None

NOISE_SHRINK = 0.005  # reference diffusion_modules.py:96,132,174


class DiffusionUtils:
    """Gaussian diffusion helper: beta schedule, forward noising, reverse sampling with optional guidance."""

    def __init__(self, n_timesteps: int = 1000, beta_min: float = 1e-4, beta_max: float = 0.02, device: str = 'cpu',
                 scheduler: str = 'linear', img_size: int = 64, data_scaled: bool = False):
        assert scheduler in ['linear', 'cosine'], 'scheduler must be linear or cosine'
        self.n_timesteps = n_timesteps
        self.beta_min = beta_min
        self.beta_max = beta_max
        self.device = device
        self.scheduler = scheduler
        self.img_size = img_size
        self.data_scaled = data_scaled
        self.betas = self.betaSamples().to(self.device)
        self.alphas = 1 - self.betas
        self.alpha_hat = torch.cumprod(self.alphas, dim=0)
        # sampling options of this implementation (the reference has none)
        self.show_progress = False
        self.noise_seed = None     # None: derive the Philox seed from torch's generator at each sample() call
        self._host_noise = None    # set by sample_with_host_noise()
        self._start_image = None
        self.first_image = 0       # index of image 0 in the global batch when this process samples one slice of it

    def betaSamples(self):
        """Linear schedule, or beta_t = 1 - abar_t / abar_{t-1} with abar = cos^2(((t/T + 0.008) / 1.008) * pi/2),
        clipped to [1e-4, 0.9999] (reference diffusion_modules.py:58-69)."""
        if self.scheduler == 'linear':
            return torch.linspace(start=self.beta_min, end=self.beta_max, steps=self.n_timesteps).to(self.device)
        t = torch.linspace(0, self.n_timesteps, self.n_timesteps + 1)
        ft = torch.cos(((t / self.n_timesteps + 0.008) / 1.008) * np.pi / 2) ** 2
        alphat = ft / ft[0]
        betat = 1 - alphat[1:] / alphat[:-1]
        return torch.clip(betat, 0.0001, 0.9999).to(self.device)

    def sampleTimesteps(self, size: int):
        return torch.randint(low=1, high=self.n_timesteps, size=(size,)).to(self.device)

    def noiseImage(self, x: torch.Tensor, t: torch.LongTensor):
        """Forward process (reference diffusion_modules.py:77-99): returns (noisy image, the noise that was added)."""
        assert len(x.shape) == 4, f'x must be a 4D tensor but is {len(x.shape)}D tensor.'
        keep = torch.sqrt(self.alpha_hat[t])[:, None, None, None]
        spread = torch.sqrt(1 - self.alpha_hat[t])[:, None, None, None]
        noise = torch.randn_like(x).to(self.device)
        if self.data_scaled:
            noise *= NOISE_SHRINK
        return (keep * x) + (spread * noise), noise

    # ------------------------------------------------------------------ reverse process
    def sample(self, n: int, model: nn.Module, channels_hr: int, y: torch.Tensor = None, cond_img: torch.Tensor = None,
               lsm_cond: torch.Tensor = None, topo_cond: torch.Tensor = None, cfg_scale: float = 0.0):
        model.eval()
        with torch.no_grad():
            if self._start_image is not None:
                x = self._start_image.to(self.device)
            else:  # drawn on the host, as the reference does, then moved
                x = torch.randn((n, channels_hr, self.img_size, self.img_size)).to(self.device)
            if self.data_scaled:
                x = x * NOISE_SHRINK
            cond_img, lsm_cond, topo_cond, y = (None if v is None else v.to(self.device)
                                                for v in (cond_img, lsm_cond, topo_cond, y))
            if isinstance(model, _HipCapableNet) and model._hip_eligible(x):
                x = self._sample_hip(x, model, y, cond_img, lsm_cond, topo_cond, cfg_scale)
            else:
                x = self._sample_generic(x, model, y, cond_img, lsm_cond, topo_cond, cfg_scale)
        model.train()  # reference diffusion_modules.py:182
        return x

    def sample_with_host_noise(self, n, model, channels_hr, noise, start_image=None, y=None, cond_img=None,
                               lsm_cond=None, topo_cond=None, cfg_scale=0.0):
        """``sample()`` with the start image and the per-step Gaussian draws supplied by the caller (before any
        ``data_scaled`` shrinking): ``noise[k]`` is used at timestep index ``n_timesteps-1-k``.  Makes a run
        comparable with a CPU run of the reference, whose draws come from the host generator."""
        self._host_noise, self._start_image = noise, start_image
        try:
            return self.sample(n, model, channels_hr, y, cond_img, lsm_cond, topo_cond, cfg_scale)
        finally:
            self._host_noise, self._start_image = None, None

    def _philox_seed(self):
        if self.noise_seed is not None:
            return int(self.noise_seed)
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def _sample_hip(self, x, model, y, cond_img, lsm_cond, topo_cond, cfg_scale):
        engine = model.hip_engine(x.device)
        x = x.detach().clone().contiguous()
        host_noise = self._host_noise
        if host_noise is not None:
            host_noise = host_noise.to(device=x.device, dtype=torch.float32)
            assert host_noise.shape[0] >= self.n_timesteps - 2, 'need one noise tensor per step with i > 1'
        args = model.hip_call_arguments(y, cond_img, lsm_cond, topo_cond)
        if cfg_scale > 0:  # the reference's second call passes nothing but x and t (diffusion_modules.py:158)
            model.hip_call_arguments(None, None, None, None)
        return engine.sample(x, *args, self.betas, self.alphas, self.alpha_hat, noise=host_noise,
                             seed=self._philox_seed(), cfg_scale=float(cfg_scale),
                             noise_scale=NOISE_SHRINK if self.data_scaled else 1.0, first_image=self.first_image)

    def _sample_generic(self, x, model, y, cond_img, lsm_cond, topo_cond, cfg_scale):
        steps = reversed(range(1, self.n_timesteps))
        if self.show_progress and _tqdm is not None:
            steps = _tqdm.tqdm(steps, position=0, total=self.n_timesteps - 1)
        draw = 0
        for i in steps:
            t = (torch.ones(x.shape[0]) * i).long().to(self.device)
            predicted_noise = model(x, t, y, cond_img, lsm_cond, topo_cond)
            if cfg_scale > 0:
                uncond = model(x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None)
                predicted_noise = torch.lerp(uncond, predicted_noise, cfg_scale)
            alpha = self.alphas[t][:, None, None, None]
            beta = self.betas[t][:, None, None, None]
            alpha_hat = self.alpha_hat[t][:, None, None, None]
            if i > 1:
                if self._host_noise is not None:
                    noise = self._host_noise[draw].to(x.device).clone()
                    draw += 1
                else:
                    noise = torch.randn_like(x).to(self.device)
                if self.data_scaled:
                    noise *= NOISE_SHRINK
            else:
                noise = torch.zeros_like(x).to(self.device)
            x = (1 / torch.sqrt(alpha)) * (x - ((1 - alpha) / torch.sqrt(1 - alpha_hat)) * predicted_noise)
            x = x + (torch.sqrt(beta) * noise)
        return x
