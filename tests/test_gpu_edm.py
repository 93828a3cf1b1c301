"""The elucidated (EDM) sampler steps in libdanra_hip.so against the oracle's restatement of
CascadedDDPM/elucidated_imagen.py:309-545.  With caller-supplied draws every kernel evaluates the reference's fp32
statements one rounding at a time, so the bar is bit equality (the toy network of tests/edm_util.py rounds alike on
both devices); the dynamic threshold is held to 1e-6 relative on the interpolated quantile and to bit equality where
the rank is an integer.  Parity with the reference itself is unpinned for this path (see oracle/edm_ref.py)."""
from math import sqrt

import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from diffusionmodelscustom_amd.elucidated_imagen import ElucidatedImagen
from oracle import edm_ref
from tests.edm_util import Tape, ToyUnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make(n_unets=1, sizes=(16, 32), **kw):
    unets = [ToyUnet(lowres_cond=i > 0, self_cond=(i == 0)) for i in range(n_unets)]
    kw.setdefault("num_sample_steps", 8)
    return ElucidatedImagen(unets, image_sizes=list(sizes)[:n_unets], channels=3, condition_on_text=False, **kw).to(DEV)


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (3, 1, 5, 7), (1, 3, 64, 64)])
def test_step_kernels_bit_equal_the_reference_statements(shape):
    torch.manual_seed(sum(shape))
    B = shape[0]
    images, z, net, inp = (torch.randn(shape) for _ in range(4))
    mask = torch.rand(B, 1, *shape[-2:]) > 0.6
    s_noise, sigma, gamma, sigma_next, sd = 1.003, 3.7, 0.4142135679721832, 2.9, 0.5
    sigma_hat = sigma + gamma * sigma
    k = sqrt(sigma_hat ** 2 - sigma ** 2)
    st = torch.full((1,), sigma_hat)
    c_in, c_skip, c_out = (f(sd, st).item() for f in (edm_ref.c_in, edm_ref.c_skip, edm_ref.c_out))
    g = lambda t: t.to(DEV).contiguous()  # noqa: E731
    new = lambda: torch.empty(shape, device=DEV)  # noqa: E731

    # churn, with and without inpainting
    eps = s_noise * z
    added = k * eps
    for masked in (False, True):
        want_hat = images + added
        if masked:
            want_hat = want_hat * ~mask + (inp + added) * mask
        want_in = edm_ref.c_in(sd, st.reshape(1, 1, 1, 1)) * want_hat
        hat, net_in = new(), new()
        hip_engine.edm_perturb(g(images), hat, net_in, s_noise, k, c_in, noise=g(z), inpaint=g(inp) if masked else None,
                               mask=g(mask.to(torch.uint8)) if masked else None)
        assert torch.equal(hat.cpu(), want_hat) and torch.equal(net_in.cpu(), want_in)

    # preconditioned output, static clamp, Euler step, next input
    hat = images + added
    padded = st.reshape(1, 1, 1, 1)
    raw = edm_ref.c_skip(sd, padded) * hat + edm_ref.c_out(sd, padded) * net
    for clamp in (False, True):
        want_out = raw.clamp(-1., 1.) if clamp else raw
        out = new()
        hip_engine.edm_precondition(g(hat), g(net), out, c_skip, c_out, clamp)
        assert torch.equal(out.cpu(), want_out)
        want_d = (hat - want_out) / sigma_hat
        want_next = hat + (sigma_next - sigma_hat) * want_d
        c_in2 = edm_ref.c_in(sd, torch.full((1,), sigma_next)).item()
        want_in2 = edm_ref.c_in(sd, torch.full((1, 1, 1, 1), sigma_next)) * want_next
        out2, d, nxt, in2 = new(), new(), new(), new()
        hip_engine.edm_euler(g(hat), g(net), out2, None, d, nxt, in2, c_skip, c_out, clamp, sigma_hat, sigma_next - sigma_hat,
                             c_in2)
        for got, want in ((out2, want_out), (d, want_d), (nxt, want_next), (in2, want_in2)):
            assert torch.equal(got.cpu(), want)

    # dynamic threshold applied by the step kernels to a model output that is already in place
    s = torch.rand(B) + 1.
    thr_out = raw.clamp(-s.reshape(-1, 1, 1, 1), s.reshape(-1, 1, 1, 1)) / s.reshape(-1, 1, 1, 1)
    out3, d, nxt = g(raw).clone(), new(), new()
    hip_engine.edm_euler(g(hat), None, out3, g(s), d, nxt, None, 0., 0., False, sigma_hat, sigma_next - sigma_hat, 0.)
    want_d = (hat - thr_out) / sigma_hat
    assert torch.equal(out3.cpu(), thr_out) and torch.equal(d.cpu(), want_d)
    assert torch.equal(nxt.cpu(), hat + (sigma_next - sigma_hat) * want_d)

    # second-order correction
    st2 = torch.full((1, 1, 1, 1), sigma_next)
    net2 = torch.randn(shape)
    want_next = hat + (sigma_next - sigma_hat) * want_d
    out_next = edm_ref.c_skip(sd, st2) * want_next + edm_ref.c_out(sd, st2) * net2
    d_prime = (want_next - out_next) / sigma_next
    want_images = hat + 0.5 * (sigma_next - sigma_hat) * (want_d + d_prime)
    got_images, out4 = new(), new()
    hip_engine.edm_heun(g(hat), g(want_next), g(net2), out4, None, g(want_d), got_images, edm_ref.c_skip(sd, st2).item(),
                        edm_ref.c_out(sd, st2).item(), False, sigma_next, 0.5 * (sigma_next - sigma_hat))
    assert torch.equal(out4.cpu(), out_next) and torch.equal(got_images.cpu(), want_images)

    # repaint noise, end of the stage
    x = g(images).clone()
    hip_engine.edm_renoise(x, sigma - sigma_next, noise=g(z))
    assert torch.equal(x.cpu(), images + (sigma - sigma_next) * z)
    for masked in (False, True):
        for unnorm in (False, True):
            want = (images * 0.7).clamp(-1., 1.)
            if masked:
                want = want * ~mask + inp * mask
            if unnorm:
                want = (want + 1.) / 2.
            x = g(images * 0.7).clone()
            hip_engine.edm_finish(x, unnorm, inpaint=g(inp) if masked else None,
                                  mask=g(mask.to(torch.uint8)) if masked else None)
            assert torch.equal(x.cpu(), want)


@pytest.mark.parametrize("n", [1, 7, 105, 768, 12288, 32768, 33333, 196608])
def test_dynamic_threshold_is_the_exact_order_statistic(n):
    torch.manual_seed(n)
    rows = torch.randn(5, n) * 3
    rows[1] = (rows[1] * 2).round() / 2          # long runs of equal values
    rows[2] = 0.25                               # one value only
    rows[3, : max(1, n // 50)] = 1e30            # a few huge values
    rows[4] = rows[4] * 0.01                     # everything below the floor of 1
    x = rows.to(DEV)
    for q in (0.0, 0.5, 0.95, 0.999, 1.0):
        got = hip_engine.edm_dynamic_threshold(x, q).cpu()
        want = torch.quantile(rows.abs(), q, dim=-1).clamp_(min=1.)
        assert torch.allclose(got, want, rtol=1e-6, atol=0), (n, q, got, want)
        rank = torch.tensor(q) * (n - 1)
        if rank == rank.floor():
            assert torch.equal(got, want), (n, q)
    assert torch.equal(got, hip_engine.edm_dynamic_threshold(x, 1.0).cpu())  # repeatable
    if n > 1:
        rows[0, n // 2] = float("nan")
        got = hip_engine.edm_dynamic_threshold(rows.to(DEV), 0.95).cpu()
        assert torch.isnan(got[0]) and not torch.isnan(got[1:]).any()


def test_refusals():
    x = torch.zeros(2, 3, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match="come together"):
        hip_engine.edm_finish(x, True, inpaint=x.clone())
    with pytest.raises(RuntimeError, match="outside"):
        hip_engine.edm_dynamic_threshold(x, 1.5)
    with pytest.raises(RuntimeError, match="sigma_next is zero"):
        hip_engine.edm_heun(x, x, x, x.clone(), None, x, x.clone(), 1., 1., False, 0., 0.)
    with pytest.raises(RuntimeError, match="contiguous CUDA float32"):
        hip_engine.edm_renoise(x.double(), 1.0, noise=x)


def oracle_run(model, shape, tape, **kw):
    unet = model.unets[0]
    dyn = kw.pop("dynamic_threshold", True)
    return edm_ref.one_unet_sample(unet.forward_with_cond_scale, shape, model.hparams[0], tape, self_cond_net=unet.self_cond,
                                   dynamic_threshold=dyn, percentile=model.dynamic_thresholding_percentile, cond_scale=1., **kw)


@pytest.mark.parametrize("case", ["static", "no_clamp", "dynamic", "inpaint", "odd_shape"])
def test_one_unet_sample_equals_the_oracle(case):
    model = make()
    shape = (3, 3, 5, 7) if case == "odd_shape" else (4, 3, 16, 16)
    kw = dict(static=dict(dynamic_threshold=False), no_clamp=dict(clamp=False, skip_steps=1, sigma_max=30.),
              dynamic=dict(), odd_shape=dict(dynamic_threshold=False),
              inpaint=dict(dynamic_threshold=False, inpaint_resample_times=3))[case]
    if case == "inpaint":
        torch.manual_seed(5)
        kw["inpaint_images"] = torch.rand(shape)
        kw["inpaint_masks"] = torch.rand(shape[0], *shape[-2:]) > 0.5
    tape = Tape(11)
    model.noise_source = tape
    dev_kw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    calls = model.unets[0].calls
    got = model.one_unet_sample(model.unets[0], shape, unet_number=1, **dev_kw).cpu()
    n_calls = model.unets[0].calls - calls
    tape.rewind()
    cpu_model = make().cpu()
    want, trace = oracle_run(cpu_model, shape, tape, **kw)
    assert n_calls == len(trace)
    if case == "dynamic":  # the interpolated quantile may differ in its last bit
        assert torch.allclose(got, want, rtol=0, atol=2e-6)
    else:
        assert torch.equal(got, want)


def test_cascade_on_the_gpu_matches_the_oracle():
    model = make(2)
    tape = Tape(13)
    model.noise_source = tape
    outs = [o.cpu() for o in model.sample(batch_size=2, return_all_unet_outputs=True)]
    n_draws = tape.cursor
    tape.rewind()
    cpu_model = make(2).cpu()
    wants = edm_ref.cascade_sample([u.forward_with_cond_scale for u in cpu_model.unets], [16, 32], cpu_model.hparams, tape,
                                   batch_size=2, self_cond=(True, False),
                                   percentile=cpu_model.dynamic_thresholding_percentile)
    assert tape.cursor == n_draws
    assert [tuple(o.shape) for o in outs] == [(2, 3, 16, 16), (2, 3, 32, 32)]
    assert torch.allclose(outs[0], wants[0], rtol=0, atol=2e-6)
    # the low-resolution conditioning image is noised with device transcendentals (log, expm1, sigmoid)
    assert torch.allclose(outs[1], wants[1], rtol=0, atol=1e-5)


def test_preconditioned_forward_equals_the_oracle():
    model = make()
    torch.manual_seed(3)
    x = torch.randn(3, 3, 16, 16)
    cpu_net = ToyUnet().forward_with_cond_scale
    for clamp, dyn in ((False, True), (True, False), (True, True)):
        want = edm_ref.preconditioned_network_forward(cpu_net, x, 2.5, sigma_data=0.5, clamp=clamp, dynamic_threshold=dyn,
                                                      percentile=model.dynamic_thresholding_percentile, cond_scale=1.)
        got = model.preconditioned_network_forward(model.unets[0].forward_with_cond_scale, x.to(DEV), 2.5, sigma_data=0.5,
                                                   clamp=clamp, dynamic_threshold=dyn, cond_scale=1.).cpu()
        assert torch.allclose(got, want, rtol=0, atol=2e-6), (clamp, dyn)
    with pytest.raises(NotImplementedError, match="one noise level"):
        model.preconditioned_network_forward(model.unets[0].forward_with_cond_scale, x.to(DEV),
                                             torch.tensor([1.0, 2.0, 3.0], device=DEV), sigma_data=0.5)


def test_in_kernel_noise_is_standard_normal_and_seeded():
    model = make(num_sample_steps=4)
    model.noise_seed = 1234
    a = model.one_unet_sample(model.unets[0], (8, 3, 64, 64), unet_number=1, clamp=False)
    b = model.one_unet_sample(model.unets[0], (8, 3, 64, 64), unet_number=1, clamp=False)
    assert torch.equal(a, b)
    model.noise_seed = 1235
    c = model.one_unet_sample(model.unets[0], (8, 3, 64, 64), unet_number=1, clamp=False)
    assert not torch.equal(a, c)
    z = torch.zeros(64, 3, 64, 64, device=DEV)
    hip_engine.edm_renoise(z, 1.0, seed=99, offset=1)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3
    assert abs((z ** 4).mean().item() - 3.0) < 5e-2
    odd = torch.zeros(2, 3, 5, 7, device=DEV)   # scalar launches take the same element -> draw mapping
    hip_engine.edm_renoise(odd, 1.0, seed=99, offset=1)
    assert torch.equal(odd.flatten(), z.flatten()[: odd.numel()])
