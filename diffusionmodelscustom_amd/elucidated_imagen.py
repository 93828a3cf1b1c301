"""The elucidated (Karras et al., EDM) sampler of the cascaded model with the call surface of the reference's
``ElucidatedImagen`` (CascadedDDPM/elucidated_imagen.py:77-751): noise-level schedule, preconditioning, static or
dynamic thresholding, the stochastic Euler step with second-order correction, inpainting with resampling, and the
cascade loop that feeds each stage's output to the next stage as its noised low-resolution conditioning image.

Scope: sampling of images.  The networks are the caller's modules (anything with the reference U-Net's
``forward_with_cond_scale(x, noise_cond, *, cond_scale, self_cond, text_embeds, text_mask, cond_images,
lowres_cond_img, lowres_noise_times)`` and its ``lowres_cond`` / ``self_cond`` attributes) and run as they are; what
runs in ``libdanra_hip.so`` is every statement of the sampling step around them.  Text conditioning enters as
precomputed ``text_embeds`` (the T5 encoder is not part of this package), video and training are not covered.

``one_unet_sample`` drives the HIP kernels on CUDA float32 tensors and nothing else: it raises if the library is
missing and it raises on a CPU device (there is no software sampler in this package; the CPU statement of the loop
lives with the tests, as their checker).  All scalars of a step are formed on the host on one-element CPU tensors /
Python floats, so that they round as in the reference's CPU run.
"""
from collections import namedtuple
from math import sqrt

import torch
import torch.nn.functional as F
from torch import nn

from . import hip_engine

Hparams = namedtuple('Hparams', ['num_sample_steps', 'sigma_min', 'sigma_max', 'sigma_data', 'rho', 'P_mean', 'P_std',
                                 'S_churn', 'S_tmin', 'S_tmax', 'S_noise'])


def _tuple(value, length):
    if isinstance(value, list):
        value = tuple(value)
    out = value if isinstance(value, tuple) else (value,) * length
    assert len(out) == length, f'expected {length} values, got {len(out)}'
    return out


def _pad_right(x, t):
    """(b,) -> (b, 1, ..., 1) with as many dimensions as x (imagen_pytorch.py:388)."""
    return t.reshape(t.shape + (1,) * (x.dim() - t.dim())) if x.dim() > t.dim() else t


def resize_image_to(image, target_image_size, clamp_range=None, mode='nearest'):
    """imagen_pytorch.py:427-451."""
    if image.shape[-1] == target_image_size:
        return image
    out = F.interpolate(image, target_image_size, mode=mode)
    return out.clamp(*clamp_range) if clamp_range is not None else out


class GaussianDiffusionContinuousTimes(nn.Module):
    """The continuous-time schedule that noises the low-resolution conditioning image (imagen_pytorch.py:549-590,
    :591-635, :724-751): log-SNR of a time in [0, 1], and the forward sample at that time."""

    def __init__(self, *, noise_schedule, timesteps=1000):
        super().__init__()
        if noise_schedule not in ('linear', 'cosine'):
            raise ValueError(f'invalid noise schedule {noise_schedule}')
        self.noise_schedule = noise_schedule
        self.num_timesteps = timesteps

    def log_snr(self, t):
        if self.noise_schedule == 'linear':
            return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))
        s = 0.008
        inner = (torch.cos((t + s) / (1 + s) * torch.pi * 0.5) ** -2) - 1
        return -torch.log(inner.clamp(min=1e-5))

    def get_times(self, batch_size, noise_level, *, device):
        return torch.full((batch_size,), noise_level, device=device, dtype=torch.float32)

    def q_sample(self, x_start, t, noise=None):
        if isinstance(t, float):
            t = torch.full((x_start.shape[0],), t, device=x_start.device, dtype=x_start.dtype)
        if noise is None:
            noise = torch.randn_like(x_start)
        log_snr = self.log_snr(t).type(x_start.dtype)
        padded = _pad_right(x_start, log_snr)
        alpha, sigma = torch.sqrt(torch.sigmoid(padded)), torch.sqrt(torch.sigmoid(-padded))
        return alpha * x_start + sigma * noise, log_snr, alpha, sigma


class ElucidatedImagen(nn.Module):
    def __init__(self, unets, *, image_sizes, text_embed_dim=None, channels=3, cond_drop_prob=0.1, resize_mode='nearest',
                 lowres_sample_noise_level=0.2, condition_on_text=True, auto_normalize_img=True, dynamic_thresholding=True,
                 dynamic_thresholding_percentile=0.95, lowres_noise_schedule='linear', num_sample_steps=32,
                 sigma_min=0.002, sigma_max=80, sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2, S_churn=80, S_tmin=0.05,
                 S_tmax=50, S_noise=1.003):
        super().__init__()
        unets = tuple(unets) if isinstance(unets, (list, tuple, nn.ModuleList)) else (unets,)
        n = len(unets)
        self.condition_on_text = condition_on_text
        self.unconditional = not condition_on_text
        self.text_embed_dim = text_embed_dim
        self.channels = channels
        self.unets = nn.ModuleList(unets)
        self.image_sizes = _tuple(image_sizes, n)
        self.sample_channels = _tuple(channels, n)
        lowres = tuple(bool(getattr(u, 'lowres_cond', False)) for u in self.unets)
        assert lowres == (False,) + (True,) * (n - 1), \
            'the first unet must be unconditioned (by low resolution image), and the rest must have `lowres_cond` set'
        self.lowres_noise_schedule = GaussianDiffusionContinuousTimes(noise_schedule=lowres_noise_schedule)
        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.resize_mode = resize_mode
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.auto_normalize_img = auto_normalize_img
        self.input_image_range = (0. if auto_normalize_img else -1., 1.)
        self.dynamic_thresholding = _tuple(dynamic_thresholding, n)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        per_unet = [_tuple(hp, n) for hp in (num_sample_steps, sigma_min, sigma_max, sigma_data, rho, P_mean, P_std, S_churn,
                                              S_tmin, S_tmax, S_noise)]
        self.hparams = [Hparams(*values) for values in zip(*per_unet)]
        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)
        # options of this implementation (the reference has none)
        self.noise_seed = None     # None: a Philox seed is drawn from torch's generator at each one_unet_sample() call
        self.noise_source = None   # callable(shape, device) -> standard normal tensor: replaces every draw, in call order
        params = list(self.unets.parameters())
        if params:
            self.to(params[0].device)

    # ---- small helpers of the reference's surface
    @property
    def device(self):
        return self._temp.device

    def normalize_img(self, img):
        return img * 2. - 1. if self.auto_normalize_img else img

    def unnormalize_img(self, img):
        return (img + 1.) / 2. if self.auto_normalize_img else img

    def resize_to(self, image, size):
        return resize_image_to(image, size, mode=self.resize_mode)

    def right_pad_dims_to_datatype(self, t):
        return t.reshape(-1, 1, 1, 1)

    # ---- thresholding and preconditioning (elucidated_imagen.py:309-369)
    def threshold_x_start(self, x_start, dynamic_threshold=True):
        if not dynamic_threshold:
            return x_start.clamp(-1., 1.)
        _require_native(x_start, 'threshold_x_start')
        s = _pad_right(x_start, hip_engine.edm_dynamic_threshold(x_start.contiguous(), self.dynamic_thresholding_percentile))
        return x_start.clamp(-s, s) / s

    def c_skip(self, sigma_data, sigma):
        return (sigma_data ** 2) / (sigma ** 2 + sigma_data ** 2)

    def c_out(self, sigma_data, sigma):
        return sigma * sigma_data * (sigma_data ** 2 + sigma ** 2) ** -0.5

    def c_in(self, sigma_data, sigma):
        return 1 * (sigma ** 2 + sigma_data ** 2) ** -0.5

    def c_noise(self, sigma):
        return torch.log(sigma.clamp(min=1e-20)) * 0.25

    def preconditioned_network_forward(self, unet_forward, noised_images, sigma, *, sigma_data, clamp=False,
                                       dynamic_threshold=True, **kwargs):
        """One denoiser evaluation at ONE noise level for the whole batch (a float, or a tensor whose entries are
        equal): input scaling, network, output combination in the library's kernel, threshold."""
        _require_native(noised_images, 'preconditioned_network_forward')
        level = float(sigma if isinstance(sigma, float) else sigma.flatten()[0])
        if torch.is_tensor(sigma) and not bool((sigma == sigma.flatten()[0]).all()):
            raise NotImplementedError('one noise level per call: every image of a sampling batch is at the same level')
        c_in, c_skip, c_out, c_noise = self._scalars(sigma_data, level)
        x = noised_images.contiguous()
        noise_cond = torch.full((x.shape[0],), c_noise, device=x.device, dtype=torch.float32)
        net_out = unet_forward(x * c_in, noise_cond, **kwargs).contiguous()
        out = torch.empty_like(x)
        hip_engine.edm_precondition(x, net_out, out, c_skip, c_out, bool(clamp and not dynamic_threshold))
        return self.threshold_x_start(out, True) if (clamp and dynamic_threshold) else out

    # ---- schedule (elucidated_imagen.py:376-390, :430-436)
    def sample_schedule(self, num_sample_steps, rho, sigma_min, sigma_max, device=None):
        """Noise levels sigma_0 > ... > sigma_{N-1}, then 0.  Evaluated on the CPU whatever the sampling device: the
        levels then round as in the reference's CPU run on every machine."""
        N = num_sample_steps
        inv_rho = 1 / rho
        steps = torch.arange(num_sample_steps, dtype=torch.float32)
        sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
        sigmas = F.pad(sigmas, (0, 1), value=0.)
        return sigmas if device is None else sigmas.to(device)

    def _levels(self, hp, sigma_min, sigma_max):
        """[(sigma, sigma_next, gamma)] as Python floats."""
        sigmas = self.sample_schedule(hp.num_sample_steps, hp.rho, sigma_min, sigma_max)
        gammas = torch.where((sigmas >= hp.S_tmin) & (sigmas <= hp.S_tmax),
                             min(hp.S_churn / hp.num_sample_steps, sqrt(2) - 1), 0.)
        return sigmas, [(s.item(), sn.item(), g.item()) for s, sn, g in zip(sigmas[:-1], sigmas[1:], gammas[:-1])]

    def _scalars(self, sigma_data, sigma):
        """c_in, c_skip, c_out, c_noise of one noise level: the reference's tensor expressions on a one-element CPU
        float32 tensor (every image of a batch is at the same level while sampling)."""
        s = torch.full((1,), sigma)
        return (self.c_in(sigma_data, s).item(), self.c_skip(sigma_data, s).item(), self.c_out(sigma_data, s).item(),
                self.c_noise(s).item())

    # ---- one stage (elucidated_imagen.py:392-545)
    @torch.no_grad()
    def one_unet_sample(self, unet, shape, *, unet_number, clamp=True, dynamic_threshold=True, cond_scale=1.,
                        use_tqdm=True, inpaint_videos=None, inpaint_images=None, inpaint_masks=None,
                        inpaint_resample_times=5, init_images=None, skip_steps=None, sigma_min=None, sigma_max=None,
                        **kwargs):
        if len(shape) != 4:
            raise NotImplementedError('video sampling (5-D shapes) is not covered')
        hp = self.hparams[unet_number - 1]
        sigma_min = hp.sigma_min if sigma_min is None else sigma_min
        sigma_max = hp.sigma_max if sigma_max is None else sigma_max
        sigmas, levels = self._levels(hp, sigma_min, sigma_max)
        levels = levels[(skip_steps or 0):]
        device = self.device
        draws = _Draws(self, device)

        inpaint_images = inpaint_videos if inpaint_videos is not None else inpaint_images
        has_inpainting = inpaint_images is not None and inpaint_masks is not None
        resample_times = inpaint_resample_times if has_inpainting else 1
        if has_inpainting:
            inpaint_images = self.resize_to(self.normalize_img(inpaint_images), shape[-1])
            inpaint_masks = self.resize_to(inpaint_masks.unsqueeze(1).float(), shape[-1]).bool()

        state = dict(unet=unet, shape=tuple(shape), hp=hp, init_sigma=sigmas[0], levels=levels, clamp=clamp,
                     dynamic_threshold=dynamic_threshold, init_images=init_images, inpaint_images=inpaint_images,
                     inpaint_masks=inpaint_masks if has_inpainting else None, resample_times=resample_times,
                     net_kwargs=dict(cond_scale=cond_scale, **kwargs), draws=draws)
        if device.type != 'cuda':
            raise NotImplementedError('the elucidated sampler of this package runs in libdanra_hip.so on a CUDA device; '
                                      f'the model is on {device}')
        return self._one_unet_sample_hip(**state)

    def _one_unet_sample_hip(self, *, unet, shape, hp, init_sigma, levels, clamp, dynamic_threshold, init_images,
                             inpaint_images, inpaint_masks, resample_times, net_kwargs, draws):
        """The same loop with every statement outside the network in the HIP library: per evaluation one kernel in front
        of the network (noise, inpaint merge, input scaling) and one behind it (output preconditioning, threshold, step),
        plus two small ones when the threshold is dynamic."""
        hip_engine.load_library()  # raises when the library was not built: no software path on a GPU
        device = self.device
        B = shape[0]
        new = lambda: torch.empty(shape, device=device, dtype=torch.float32)  # noqa: E731
        images, images_hat, images_next, net_in, d = new(), new(), new(), new(), new()
        out_first, out_second = new(), new()
        thr = torch.empty(B, device=device, dtype=torch.float32)
        uses_self_cond = bool(getattr(unet, 'self_cond', False))
        has_inpainting = inpaint_masks is not None
        inpaint = mask = None
        if has_inpainting:
            inpaint = inpaint_images.to(device=device, dtype=torch.float32).expand(shape).contiguous()
            mask = inpaint_masks.to(device=device).expand(B, 1, *shape[-2:]).to(torch.uint8).contiguous()
        dynamic = bool(clamp and dynamic_threshold)
        static_clamp = bool(clamp and not dynamic_threshold)
        pct = self.dynamic_thresholding_percentile

        images.zero_()
        hip_engine.edm_renoise(images, init_sigma.item(), **draws.kernel_args(shape))
        if init_images is not None:
            images += init_images
        x_start = None

        def evaluate(x_in, c_noise, self_cond):
            noise_cond = torch.full((B,), c_noise, device=device, dtype=torch.float32)
            net_out = unet.forward_with_cond_scale(x_in, noise_cond, self_cond=self_cond, **net_kwargs)
            if not (net_out.dtype == torch.float32 and net_out.shape == x_in.shape):
                raise RuntimeError(f'the network returned {tuple(net_out.shape)} {net_out.dtype}; '
                                   f'expected {tuple(x_in.shape)} float32')
            return net_out.contiguous()

        for ind, (sigma, sigma_next, gamma) in enumerate(levels):
            is_last_timestep = ind == len(levels) - 1
            sigma_hat = sigma + gamma * sigma
            c_in, c_skip, c_out, c_noise = self._scalars(hp.sigma_data, sigma_hat)
            second_order = sigma_next != 0
            if second_order:
                c_in2, c_skip2, c_out2, c_noise2 = self._scalars(hp.sigma_data, sigma_next)
            for r in reversed(range(resample_times)):
                hip_engine.edm_perturb(images, images_hat, net_in, hp.S_noise, sqrt(sigma_hat ** 2 - sigma ** 2), c_in,
                                       inpaint=inpaint, mask=mask, **draws.kernel_args(shape))
                net_out = evaluate(net_in, c_noise, x_start if uses_self_cond else None)
                if dynamic:
                    hip_engine.edm_precondition(images_hat, net_out, out_first, c_skip, c_out, False)
                    hip_engine.edm_dynamic_threshold(out_first, pct, out=thr)
                    net_out = None
                hip_engine.edm_euler(images_hat, net_out, out_first, thr if dynamic else None, d,
                                     images_next if second_order else images, net_in if second_order else None,
                                     c_skip, c_out, static_clamp, sigma_hat, sigma_next - sigma_hat,
                                     c_in2 if second_order else 0.)
                if second_order:
                    net_out = evaluate(net_in, c_noise2, out_first if uses_self_cond else None)
                    if dynamic:
                        hip_engine.edm_precondition(images_next, net_out, out_second, c_skip2, c_out2, False)
                        hip_engine.edm_dynamic_threshold(out_second, pct, out=thr)
                        net_out = None
                    hip_engine.edm_heun(images_hat, images_next, net_out, out_second, thr if dynamic else None, d, images,
                                        c_skip2, c_out2, static_clamp, sigma_next, 0.5 * (sigma_next - sigma_hat))
                if has_inpainting and not (r == 0 or is_last_timestep):
                    hip_engine.edm_renoise(images, sigma - sigma_next, **draws.kernel_args(shape))
                x_start = out_second if second_order else out_first
        hip_engine.edm_finish(images, self.auto_normalize_img, inpaint=inpaint, mask=mask)
        return images

    # ---- the cascade (elucidated_imagen.py:547-751)
    @torch.no_grad()
    def sample(self, texts=None, text_masks=None, text_embeds=None, cond_images=None, inpaint_images=None,
               inpaint_masks=None, inpaint_resample_times=5, init_images=None, skip_steps=None, sigma_min=None,
               sigma_max=None, batch_size=1, cond_scale=1., lowres_sample_noise_level=None, start_at_unet_number=1,
               start_image_or_video=None, stop_at_unet_number=None, return_all_unet_outputs=False, use_tqdm=True,
               device=None):
        was_training = self.training
        self.eval()
        try:
            return self._sample(texts, text_masks, text_embeds, cond_images, inpaint_images, inpaint_masks,
                                inpaint_resample_times, init_images, skip_steps, sigma_min, sigma_max, batch_size,
                                cond_scale, lowres_sample_noise_level, start_at_unet_number, start_image_or_video,
                                stop_at_unet_number, return_all_unet_outputs, use_tqdm, device)
        finally:
            self.train(was_training)

    def _sample(self, texts, text_masks, text_embeds, cond_images, inpaint_images, inpaint_masks, inpaint_resample_times,
                init_images, skip_steps, sigma_min, sigma_max, batch_size, cond_scale, lowres_sample_noise_level,
                start_at_unet_number, start_image_or_video, stop_at_unet_number, return_all_unet_outputs, use_tqdm, device):
        if device is not None:
            self.to(device)
        device = self.device
        if cond_images is not None and cond_images.dtype == torch.uint8:
            cond_images = cond_images.float() / 255
        if texts is not None and text_embeds is None and not self.unconditional:
            raise NotImplementedError('pass `text_embeds` (and `text_masks`): the text encoder is not part of this package')
        if not self.unconditional:
            assert text_embeds is not None, 'text must be passed in if the network was not trained without text'
            if text_masks is None:
                text_masks = torch.any(text_embeds != 0., dim=-1)
            batch_size = text_embeds.shape[0]
        if inpaint_images is not None:
            if self.unconditional and batch_size == 1:
                batch_size = inpaint_images.shape[0]
            assert inpaint_images.shape[0] == batch_size, \
                'number of inpainting images must be equal to the specified batch size on sample `sample(batch_size=<int>)`'
        assert not (self.condition_on_text and text_embeds is None), 'text or text encodings must be passed into imagen if specified'
        assert not (not self.condition_on_text and text_embeds is not None), 'imagen specified not to be conditioned on text, yet it is presented'
        assert not (text_embeds is not None and self.text_embed_dim is not None and text_embeds.shape[-1] != self.text_embed_dim), \
            f'invalid text embedding dimension being passed in (should be {self.text_embed_dim})'
        assert not ((inpaint_images is None) ^ (inpaint_masks is None)), 'inpaint images and masks must be both passed in to do inpainting'

        n = len(self.unets)
        if lowres_sample_noise_level is None:
            lowres_sample_noise_level = self.lowres_sample_noise_level
        cond_scale = _tuple(cond_scale, n)
        init_images = [None if im is None else self.normalize_img(im) for im in _tuple(init_images, n)]
        skip_steps, sigma_min, sigma_max = _tuple(skip_steps, n), _tuple(sigma_min, n), _tuple(sigma_max, n)
        img = None
        if start_at_unet_number > 1:
            assert start_at_unet_number <= n, 'must start a unet that is less than the total number of unets'
            assert stop_at_unet_number is None or start_at_unet_number <= stop_at_unet_number
            assert start_image_or_video is not None, 'starting image or video must be supplied if only doing upscaling'
            img = self.resize_to(start_image_or_video, self.image_sizes[start_at_unet_number - 2])

        outputs = []
        for index, unet in enumerate(self.unets):
            unet_number = index + 1
            if unet_number < start_at_unet_number:
                continue
            image_size = self.image_sizes[index]
            lowres_cond_img = lowres_noise_times = None
            if getattr(unet, 'lowres_cond', False):
                lowres_noise_times = self.lowres_noise_schedule.get_times(batch_size, lowres_sample_noise_level, device=device)
                lowres_cond_img = self.normalize_img(self.resize_to(img, image_size))
                draw = _Draws(self, device).normal(lowres_cond_img.shape)
                lowres_cond_img, *_ = self.lowres_noise_schedule.q_sample(x_start=lowres_cond_img, t=lowres_noise_times,
                                                                          noise=draw)
            stage_init = init_images[index]
            if stage_init is not None:
                stage_init = self.resize_to(stage_init, image_size)
            shape = (batch_size, self.channels, image_size, image_size)
            img = self.one_unet_sample(unet, shape, unet_number=unet_number, text_embeds=text_embeds, text_mask=text_masks,
                                       cond_images=cond_images, inpaint_images=inpaint_images, inpaint_masks=inpaint_masks,
                                       inpaint_resample_times=inpaint_resample_times, init_images=stage_init,
                                       skip_steps=skip_steps[index], sigma_min=sigma_min[index], sigma_max=sigma_max[index],
                                       cond_scale=cond_scale[index], lowres_cond_img=lowres_cond_img,
                                       lowres_noise_times=lowres_noise_times,
                                       dynamic_threshold=self.dynamic_thresholding[index], use_tqdm=use_tqdm)
            outputs.append(img)
            if stop_at_unet_number is not None and stop_at_unet_number == unet_number:
                break
        return outputs if return_all_unet_outputs else outputs[-1]


def _require_native(t, what):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() >= 2):
        raise NotImplementedError(f'{what} runs in libdanra_hip.so: it needs a CUDA float32 tensor, got '
                                  f'{t.dtype} on {t.device}')


class _Draws:
    """Where the Gaussian draws of one sampling call come from: the model's `noise_source` (every draw, in the
    reference's order: initial noise, then per step the churn noise and, while inpainting, the repaint noise), else on
    the GPU the kernels' Philox streams (one seed per call, one stream offset per draw), else torch.randn."""

    def __init__(self, model, device):
        self.source = model.noise_source
        self.device = device
        self.count = 0
        self._model_seed = model.noise_seed
        self._seed = None

    @property
    def seed(self):
        """The Philox seed of this call, drawn from torch's generator the first time a kernel asks for it (a call whose
        draws come from `noise_source` or from torch.randn leaves the generator alone)."""
        if self._seed is None:
            self._seed = int(self._model_seed) if self._model_seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        return self._seed

    def normal(self, shape):
        if self.source is not None:
            return self.source(tuple(shape), self.device).to(device=self.device, dtype=torch.float32)
        return torch.randn(shape, device=self.device)

    def kernel_args(self, shape):
        if self.source is not None:
            return dict(noise=self.normal(shape).contiguous())
        self.count += 1
        return dict(seed=self.seed, offset=self.count)
