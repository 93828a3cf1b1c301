"""Host side of the elucidated (EDM) sampler without a GPU: the package's CPU path against the oracle's restatement of
CascadedDDPM/elucidated_imagen.py (bit for bit: both evaluate torch CPU statements), the schedule's known end points,
and the oracle's spelled-out quantile against torch.quantile.  Parity with the reference itself is unpinned for this
path (see oracle/edm_ref.py)."""
import torch

from diffusionmodelscustom_amd.elucidated_imagen import ElucidatedImagen, GaussianDiffusionContinuousTimes
from oracle import edm_ref
from tests.edm_util import Tape, ToyUnet


def make(n_unets=1, **kw):
    unets = [ToyUnet(lowres_cond=i > 0, self_cond=(i == 0)) for i in range(n_unets)]
    sizes = [8, 16][:n_unets]
    kw.setdefault("num_sample_steps", 6)
    return ElucidatedImagen(unets, image_sizes=sizes, channels=3, condition_on_text=False, **kw)


def test_schedule_end_points_and_monotony():
    model = make()
    sig = model.sample_schedule(32, 7, 0.002, 80)
    assert sig.shape == (33,) and sig[-1] == 0
    assert abs(sig[0].item() - 80) < 1e-4 and abs(sig[31].item() - 0.002) < 1e-7
    assert bool((sig[:-1] > sig[1:]).all())
    assert torch.equal(sig, edm_ref.sample_schedule(32, 7, 0.002, 80))


def test_spelled_out_quantile_is_torch_quantile():
    torch.manual_seed(0)
    for n in (5, 64, 777, 12288):
        rows = torch.randn(3, n).abs()
        rows[1] = (rows[1] * 4).round() / 4  # ties
        for q in (0.0, 0.5, 0.95, 1.0):
            assert torch.equal(edm_ref.quantile_by_sorting(rows, q), torch.quantile(rows, q, dim=-1)), (n, q)


def run_pair(model, shape, **kw):
    tape = Tape(7)
    model.noise_source = tape
    got = model.one_unet_sample(model.unets[0], shape, unet_number=1, **kw)
    n_draws = tape.cursor
    tape.rewind()
    unet = model.unets[0]
    okw = dict(kw)
    dyn = okw.pop("dynamic_threshold", True)
    want, _ = edm_ref.one_unet_sample(unet.forward_with_cond_scale, shape, model.hparams[0], tape, self_cond_net=unet.self_cond,
                                      dynamic_threshold=dyn, percentile=model.dynamic_thresholding_percentile,
                                      cond_scale=1., **okw)
    assert tape.cursor == n_draws
    return got, want


def test_cpu_path_equals_the_oracle():
    model = make()
    for dyn in (True, False):
        got, want = run_pair(model, (2, 3, 8, 8), dynamic_threshold=dyn)
        assert torch.equal(got, want)
    got, want = run_pair(model, (2, 3, 8, 8), clamp=False, skip_steps=2, sigma_max=20.)
    assert torch.equal(got, want)


def test_cpu_inpainting_equals_the_oracle():
    model = make()
    torch.manual_seed(1)
    inpaint = torch.rand(2, 3, 8, 8)
    masks = torch.rand(2, 8, 8) > 0.5
    got, want = run_pair(model, (2, 3, 8, 8), inpaint_images=inpaint, inpaint_masks=masks, inpaint_resample_times=2)
    assert torch.equal(got, want)
    keep = masks.unsqueeze(1).expand_as(got)
    assert torch.allclose(got[keep], inpaint[keep], atol=1e-6)  # the masked pixels come back as they were given


def test_cascade_feeds_the_noised_upsampled_stage_to_the_next():
    model = make(2)
    tape = Tape(3)
    model.noise_source = tape
    outs = model.sample(batch_size=2, return_all_unet_outputs=True)
    assert [tuple(o.shape) for o in outs] == [(2, 3, 8, 8), (2, 3, 16, 16)]
    # the same by hand from the oracle's pieces
    tape.rewind()
    first, _ = edm_ref.one_unet_sample(model.unets[0].forward_with_cond_scale, (2, 3, 8, 8), model.hparams[0], tape,
                                       self_cond_net=True, cond_scale=1.)
    assert torch.equal(outs[0], first)
    up = torch.nn.functional.interpolate(first, 16, mode="nearest") * 2. - 1.
    times = torch.full((2,), 0.2)
    low = edm_ref.lowres_q_sample(up, times, tape((2, 3, 16, 16)))
    second, _ = edm_ref.one_unet_sample(model.unets[1].forward_with_cond_scale, (2, 3, 16, 16), model.hparams[1], tape,
                                        cond_scale=1., lowres_cond_img=low, lowres_noise_times=times)
    assert torch.equal(outs[1], second)


def test_lowres_schedule_matches_the_oracle():
    sched = GaussianDiffusionContinuousTimes(noise_schedule="linear")
    x, z, t = torch.randn(2, 3, 4, 4), torch.randn(2, 3, 4, 4), torch.tensor([0.2, 0.7])
    assert torch.equal(sched.q_sample(x, t, z)[0], edm_ref.lowres_q_sample(x, t, z))
