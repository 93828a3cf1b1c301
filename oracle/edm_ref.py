"""CPU restatement of the elucidated (EDM) sampler of the reference's cascaded model.

TEST INFRASTRUCTURE ONLY: the checker that the HIP sampler steps are compared against.  Only ``tests/`` may import it;
the product package never does.

A functional restatement, in torch CPU fp32 ops, of /root/reference/CascadedDDPM/elucidated_imagen.py:309-545
(``threshold_x_start``, the preconditioning coefficients, ``preconditioned_network_forward``, ``sample_schedule``,
``one_unet_sample``) and of the low-resolution noising it calls (imagen_pytorch.py:549-590, :724-751), each function
citing the lines it follows.  Gaussian draws are arguments (a callable that is asked for one tensor per draw, in the
reference's order), so that a run can be repeated on the device with the same draws.

PARITY UNPINNED: the reference module cannot be imported in the build container (kornia, beartype and torchvision are
not installed, and elucidated_imagen.py:40 imports ``unnormalize_zero_to_one``, a name imagen_pytorch.py does not
define), and the reference holds no test, fixture or stored output for this path.  What pins this file is its
line-by-line correspondence with the cited statements only.
"""
from math import sqrt

import torch
import torch.nn.functional as F


def threshold_x_start(x_start, dynamic_threshold=True, percentile=0.95):
    """elucidated_imagen.py:309-321."""
    if not dynamic_threshold:
        return x_start.clamp(-1., 1.)
    s = torch.quantile(x_start.reshape(x_start.shape[0], -1).abs(), percentile, dim=-1)
    s.clamp_(min=1.)
    s = s.reshape(-1, *((1,) * (x_start.dim() - 1)))
    return x_start.clamp(-s, s) / s


def quantile_by_sorting(rows, q):
    """torch.quantile(rows, q, dim=-1) spelled out (ATen's quantile_compute: sort, rank = q (n - 1) in the input's
    dtype, linear interpolation between the two neighbouring order statistics), for the order-statistic test."""
    ordered, _ = rows.sort(dim=-1)
    rank = torch.tensor(q, dtype=rows.dtype) * (rows.shape[-1] - 1)
    below = rank.floor()
    weight = rank - below
    lo = ordered[:, int(below.item())]
    hi = ordered[:, int(rank.ceil().item())]
    return lo.lerp(hi, weight)


def c_skip(sigma_data, sigma):
    """elucidated_imagen.py:325-326."""
    return (sigma_data ** 2) / (sigma ** 2 + sigma_data ** 2)


def c_out(sigma_data, sigma):
    """elucidated_imagen.py:328-329."""
    return sigma * sigma_data * (sigma_data ** 2 + sigma ** 2) ** -0.5


def c_in(sigma_data, sigma):
    """elucidated_imagen.py:331-332."""
    return 1 * (sigma ** 2 + sigma_data ** 2) ** -0.5


def c_noise(sigma):
    """elucidated_imagen.py:334-335 with the file's log helper (:72-73)."""
    return torch.log(sigma.clamp(min=1e-20)) * 0.25


def preconditioned_network_forward(net, noised_images, sigma, *, sigma_data, clamp=False, dynamic_threshold=True,
                                   percentile=0.95, **kwargs):
    """elucidated_imagen.py:340-369.  `net(x, noise_cond, **kwargs)` stands for unet.forward_with_cond_scale."""
    batch = noised_images.shape[0]
    if isinstance(sigma, float):
        sigma = torch.full((batch,), sigma)
    padded = sigma.reshape(-1, 1, 1, 1)
    net_out = net(c_in(sigma_data, padded) * noised_images, c_noise(sigma), **kwargs)
    out = c_skip(sigma_data, padded) * noised_images + c_out(sigma_data, padded) * net_out
    if not clamp:
        return out
    return threshold_x_start(out, dynamic_threshold, percentile)


def sample_schedule(num_sample_steps, rho, sigma_min, sigma_max):
    """elucidated_imagen.py:376-390."""
    N = num_sample_steps
    inv_rho = 1 / rho
    steps = torch.arange(num_sample_steps, dtype=torch.float32)
    sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
    return F.pad(sigmas, (0, 1), value=0.)


def one_unet_sample(net, shape, hp, normal, *, self_cond_net=False, clamp=True, dynamic_threshold=True, percentile=0.95,
                    inpaint_images=None, inpaint_masks=None, inpaint_resample_times=5, init_images=None, skip_steps=None,
                    sigma_min=None, sigma_max=None, auto_normalize_img=True, **kwargs):
    """elucidated_imagen.py:392-545 for images.  `hp` has the fields of the reference's Hparams; `normal(shape)` returns
    the next standard normal draw; `inpaint_masks` is [b, h, w] boolean as the caller of the reference passes it.
    Returns (images, trace) where trace lists every model output (the x_start of each evaluation) in order."""
    sigma_min = hp.sigma_min if sigma_min is None else sigma_min
    sigma_max = hp.sigma_max if sigma_max is None else sigma_max
    sigmas = sample_schedule(hp.num_sample_steps, hp.rho, sigma_min, sigma_max)
    gammas = torch.where((sigmas >= hp.S_tmin) & (sigmas <= hp.S_tmax), min(hp.S_churn / hp.num_sample_steps, sqrt(2) - 1), 0.)
    sigmas_and_gammas = list(zip(sigmas[:-1], sigmas[1:], gammas[:-1]))
    images = sigmas[0] * normal(shape)
    if init_images is not None:
        images += init_images
    x_start = None
    has_inpainting = inpaint_images is not None and inpaint_masks is not None
    resample_times = inpaint_resample_times if has_inpainting else 1
    if has_inpainting:
        if auto_normalize_img:
            inpaint_images = inpaint_images * 2. - 1.
        if inpaint_images.shape[-1] != shape[-1]:
            inpaint_images = F.interpolate(inpaint_images, shape[-1], mode='nearest')
        inpaint_masks = inpaint_masks.unsqueeze(1).float()
        if inpaint_masks.shape[-1] != shape[-1]:
            inpaint_masks = F.interpolate(inpaint_masks, shape[-1], mode='nearest')
        inpaint_masks = inpaint_masks.bool()
    unet_kwargs = dict(sigma_data=hp.sigma_data, clamp=clamp, dynamic_threshold=dynamic_threshold, percentile=percentile, **kwargs)
    sigmas_and_gammas = sigmas_and_gammas[(skip_steps or 0):]
    total_steps = len(sigmas_and_gammas)
    trace = []
    for ind, (sigma, sigma_next, gamma) in enumerate(sigmas_and_gammas):
        is_last_timestep = ind == (total_steps - 1)
        sigma, sigma_next, gamma = sigma.item(), sigma_next.item(), gamma.item()
        for r in reversed(range(resample_times)):
            is_last_resample_step = r == 0
            eps = hp.S_noise * normal(shape)
            sigma_hat = sigma + gamma * sigma
            added_noise = sqrt(sigma_hat ** 2 - sigma ** 2) * eps
            images_hat = images + added_noise
            self_cond = x_start if self_cond_net else None
            if has_inpainting:
                images_hat = images_hat * ~inpaint_masks + (inpaint_images + added_noise) * inpaint_masks
            model_output = preconditioned_network_forward(net, images_hat, sigma_hat, self_cond=self_cond, **unet_kwargs)
            trace.append(model_output)
            denoised_over_sigma = (images_hat - model_output) / sigma_hat
            images_next = images_hat + (sigma_next - sigma_hat) * denoised_over_sigma
            has_second_order_correction = sigma_next != 0
            if has_second_order_correction:
                self_cond = model_output if self_cond_net else None
                model_output_next = preconditioned_network_forward(net, images_next, sigma_next, self_cond=self_cond,
                                                                   **unet_kwargs)
                trace.append(model_output_next)
                denoised_prime_over_sigma = (images_next - model_output_next) / sigma_next
                images_next = images_hat + 0.5 * (sigma_next - sigma_hat) * (denoised_over_sigma + denoised_prime_over_sigma)
            images = images_next
            if has_inpainting and not (is_last_resample_step or is_last_timestep):
                repaint_noise = normal(shape)
                images = images + (sigma - sigma_next) * repaint_noise
            x_start = model_output if not has_second_order_correction else model_output_next
    images = images.clamp(-1., 1.)
    if has_inpainting:
        images = images * ~inpaint_masks + inpaint_images * inpaint_masks
    if auto_normalize_img:
        images = (images + 1.) / 2.  # imagen_pytorch.py:511-520 (the function elucidated_imagen.py:198 means)
    return images, trace


def beta_linear_log_snr(t):
    """imagen_pytorch.py:549-561."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def lowres_q_sample(x_start, t, noise):
    """imagen_pytorch.py:724-751 with the linear schedule and log_snr_to_alpha_sigma (:577-589)."""
    log_snr = beta_linear_log_snr(t).type(x_start.dtype).reshape(-1, 1, 1, 1)
    alpha, sigma = torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))
    return alpha * x_start + sigma * noise


def cascade_sample(nets, image_sizes, hparams, normal, *, batch_size, channels=3, lowres_sample_noise_level=0.2,
                   self_cond=(), dynamic_thresholding=True, percentile=0.95, auto_normalize_img=True):
    """elucidated_imagen.py:547-751 for unconditional images: stage k > 0 is given the previous stage's output, resized
    (nearest) to its size, normalised and noised at `lowres_sample_noise_level` (one draw), then sampled with
    one_unet_sample.  `nets[k](x, noise_cond, **kwargs)` stands for unet k's forward_with_cond_scale.  Returns every
    stage's output."""
    outputs, img = [], None
    for k, (net, size, hp) in enumerate(zip(nets, image_sizes, hparams)):
        kwargs = {}
        if k > 0:
            times = torch.full((batch_size,), lowres_sample_noise_level)
            low = img if img.shape[-1] == size else F.interpolate(img, size, mode='nearest')
            if auto_normalize_img:
                low = low * 2. - 1.
            kwargs = dict(lowres_cond_img=lowres_q_sample(low, times, normal(low.shape)), lowres_noise_times=times)
        img, _ = one_unet_sample(net, (batch_size, channels, size, size), hp, normal,
                                 self_cond_net=bool(self_cond[k]) if self_cond else False,
                                 dynamic_threshold=dynamic_thresholding, percentile=percentile,
                                 auto_normalize_img=auto_normalize_img, cond_scale=1., **kwargs)
        outputs.append(img)
    return outputs
