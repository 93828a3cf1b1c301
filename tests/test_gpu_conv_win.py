"""The window-staged 3x3 convolution kernel (csrc/kernels_conv_win.hip) through its per-op C-ABI entry, danra_conv3x3,
against a float64 evaluation of torch.nn.functional.conv2d (what torchvision's BasicBlock and the reference's
DecoderBlock apply: modules_DANRA_conditional.py:276-303, 415-416, 436-437): every tile shape the kernel has (128-, 64-
and 32-pixel tiles, pixel-major tiles of 2x2 maps, one to four 64-channel blocks per stage, one or two window
buffers), ragged batches, InstanceNorm on load, the epilogue terms, and the value ranges in which a workgroup must
leave the split-operand fp16 products for fp32 arithmetic.  Per-op tolerance: 1e-5 of the result's scale."""
import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from gpu_util import conv3x3, conv3x3s2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_OP = 1e-5


def reference(x, w, bias=None, res=None, norm=None, relu=False):
    xd = x.double()
    if norm is not None:
        xd = (x - norm[0][:, :, None, None]) * norm[1][:, :, None, None]   # fp32, as the kernel forms it on load
        xd = xd.double()
    y = torch.nn.functional.conv2d(xd, w.double(), None if bias is None else bias.double(), padding=1)
    if res is not None:
        y = y + res.double()
    return torch.relu(y) if relu else y


def run(x, w, bias=None, res=None, norm=None, relu=False):
    lib = hip_engine.load_library()
    g = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    return conv3x3(lib, g(x), g(w), g(bias), g(res), None if norm is None else (g(norm[0]), g(norm[1])), relu).cpu().double()


def rel(got, want):
    return float((got - want).abs().max() / max(float(want.abs().max()), 1e-30))


# (batch, H, W, Cin, Cout): what tile shape / staging form each one takes is noted beside it
SHAPES = [
    # 128-pixel tiles (rows of one image + the rows above and below), pixel tiles dealt to two wave groups
    (64, 16, 16, 64, 64),     # window of 160 pixels, one channel block
    (16, 32, 32, 64, 64),     # 192 pixels
    (8, 64, 64, 64, 32),      # 256 pixels: the widest map
    (64, 16, 16, 128, 64),    # two channel blocks in one stage
    (64, 16, 16, 192, 64),    # three channel blocks, one per stage
    # 64-pixel tiles, the k-steps dealt alternately to two wave groups
    (64, 8, 8, 128, 128),     # whole images, two channel blocks in one stage
    (64, 8, 8, 256, 128),     # four blocks
    (64, 16, 16, 64, 32),     # four rows + halo rows (96 pixels), one block: five and four taps per wave group
    (2, 64, 64, 64, 64),      # one 64-wide row + halo rows (192 pixels)
    # 32-pixel tiles
    (64, 4, 4, 256, 256),     # two images per tile, four channel blocks in one stage
    (64, 4, 4, 512, 256),     # two stages, two window buffers
    (8, 32, 32, 64, 32),      # one row + halo rows (96 pixels)
    (1, 16, 16, 64, 32),      # one image, two rows + halo rows
    (3, 4, 4, 256, 64),       # ragged: the last tile holds one image
    (5, 2, 2, 512, 96),       # a 2x2 map whose batch does not fill pixel-major tiles: whole-image tiles, ragged
    (2, 8, 16, 128, 64),      # a rectangular map
    (7, 1, 2, 64, 32),        # a one-row map
    # pixel-major tiles of 2x2 maps
    (64, 2, 2, 512, 512),     # four stages of two blocks
    (32, 2, 2, 128, 32),      # one stage
    (32, 2, 2, 64, 32),       # one block
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_every_tile_shape_against_float64(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H * 100 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    got = run(x, w, bias, res, relu=True)
    want = reference(x, w, bias, res, relu=True)
    assert got.shape == want.shape
    assert rel(got, want) < TOL_OP
    plain = run(x, w)                                           # no bias, no residual, no activation
    assert rel(plain, reference(x, w)) < TOL_OP
    assert torch.equal(run(x, w, bias, res, relu=True), got)    # bitwise reproducible


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(64, 16, 16, 128, 64), (16, 32, 32, 64, 64), (8, 64, 64, 64, 32), (2, 64, 64, 64, 64), (4, 16, 16, 256, 64)])
def test_instance_norm_on_load(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g) * 3.0 + 1.0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    mean = x.mean(dim=(2, 3))
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3), unbiased=False) + 1e-5)
    bias = torch.randn(Cout, generator=g)
    got = run(x, w, bias, norm=(mean, rstd))
    assert rel(got, reference(x, w, bias, norm=(mean, rstd))) < TOL_OP


def test_unsupported_shapes_are_refused():
    lib = hip_engine.load_library()
    x = torch.zeros(1, 64, 12, 12, device=DEV)
    w = torch.ones(32, 64, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="not a shape of the window-staged kernel"):
        conv3x3(lib, x, w)
    with pytest.raises(RuntimeError, match="not a shape of the window-staged kernel"):
        conv3x3(lib, torch.zeros(1, 48, 8, 8, device=DEV), torch.ones(32, 48, 3, 3, device=DEV))
    with pytest.raises(RuntimeError, match="all zero or not finite"):
        conv3x3(lib, torch.zeros(1, 64, 8, 8, device=DEV), torch.zeros(32, 64, 3, 3, device=DEV))


CASES = ["huge_value", "huge_region", "tiny", "tiny_with_zeros", "zeros", "band_gaussian_3e-3", "band_uniform_2^-8_2^-3",
         "band_mixed_with_order_one"]


@pytest.mark.parametrize("shape", [(8, 16, 16, 64, 64), (8, 4, 4, 256, 64), (64, 2, 2, 512, 32)])
@pytest.mark.parametrize("case", CASES)
def test_value_ranges(shape, case):
    """A workgroup whose staged window holds a value of 2^15 or more, or nothing of 2^-8 or more (unless it is all zero),
    forms its tile with fp32 multiply-adds; everything else multiplies in the split-operand form, including the band
    below 2^-3 where the low halves are fp16 subnormals.  Judged per image, so that one huge image does not hide the
    error of the others."""
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    if case == "huge_value":
        x[0, 5, H // 2, W - 1] = 3.0e6
    elif case == "huge_region":
        x[1, :, H // 2:, :] *= 1.0e5
    elif case == "tiny":
        x *= 1.0e-6
    elif case == "tiny_with_zeros":
        x *= 1.0e-5
        x[:, ::2] = 0.0
    elif case == "zeros":
        x[0] = 0.0
    elif case == "band_gaussian_3e-3":
        x *= 3.0e-3
    elif case == "band_uniform_2^-8_2^-3":
        x = torch.exp2(torch.rand(x.shape, generator=g) * 5.0 - 8.0) * torch.sign(x)
    elif case == "band_mixed_with_order_one":
        x[:, ::2] *= 4.0e-3
    got = run(x, w)
    want = reference(x, w)
    assert torch.isfinite(got).all()
    for b in range(B):
        scale = max(float(want[b].abs().max()), 1e-30)
        assert float((got[b] - want[b]).abs().max()) <= TOL_OP * scale, (case, b)


def test_nan_reaches_the_outputs_whose_window_holds_it_and_no_others():
    B, H, W, Cin, Cout = 4, 16, 16, 64, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    clean = run(x, w)
    x[2, 7, 5, 9] = float("nan")
    got = run(x, w)
    hit = torch.zeros(B, H, W, dtype=torch.bool)
    hit[2, 4:7, 8:11] = True
    assert torch.isnan(got.permute(0, 2, 3, 1)[hit]).all()
    keep = ~hit[:, None].expand_as(got)
    assert torch.equal(got[keep], clean[keep])


# ---- stride 2 (the first convolution of a down-sampling BasicBlock, with the block's 1x1 stride-2 shortcut as further
# output channels: torchvision resnet.BasicBlock under modules_DANRA_conditional.py:285-303)
S2_SHAPES = [
    (64, 16, 16, 64, 256),    # one output image per 64-pixel tile, one channel block
    (64, 8, 8, 128, 512),     # two images per 32-pixel tile, two channel blocks in one stage
    (64, 4, 4, 256, 1024),    # eight images per tile, two stages, two window buffers
    (3, 8, 8, 128, 64),       # ragged: the last tile holds one image
    (5, 4, 4, 256, 96),
    (1, 16, 16, 64, 32),
    (6, 2, 2, 128, 64),       # a one-pixel output map
    (2, 8, 16, 64, 64),       # a rectangular map
]


def reference_s2(x, w, bias, n_split, relu):
    y = torch.nn.functional.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride=2, padding=1)
    if not n_split:
        return torch.relu(y) if relu else y
    first = torch.relu(y[:, :n_split]) if relu else y[:, :n_split]
    return first, y[:, n_split:]


@pytest.mark.parametrize("B,H,W,Cin,Cout", S2_SHAPES)
def test_stride_two_against_float64(B, H, W, Cin, Cout):
    lib = hip_engine.load_library()
    g = torch.Generator().manual_seed(B * 1000 + H * 100 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    got = conv3x3s2(lib, x.to(DEV), w.to(DEV), bias.to(DEV), relu=True).cpu().double()
    want = reference_s2(x, w, bias, 0, True)
    assert got.shape == want.shape
    assert rel(got, want) < TOL_OP
    plain = conv3x3s2(lib, x.to(DEV), w.to(DEV)).cpu().double()
    assert rel(plain, reference_s2(x, w, None, 0, False)) < TOL_OP
    assert torch.equal(conv3x3s2(lib, x.to(DEV), w.to(DEV), bias.to(DEV), relu=True).cpu().double(), got)
    if Cout % 64 == 0:
        # the second half of the channels as a block's shortcut: zero but for the centre tap, no ReLU, its own tensor
        n_split = Cout // 2
        w2 = w.clone()
        centre = w2[n_split:, :, 1, 1].clone()
        w2[n_split:] = 0.0
        w2[n_split:, :, 1, 1] = centre
        a, b2 = conv3x3s2(lib, x.to(DEV), w2.to(DEV), bias.to(DEV), n_split=n_split, relu=True)
        wa, wb = reference_s2(x, w2, bias, n_split, True)
        assert rel(a.cpu().double(), wa) < TOL_OP and rel(b2.cpu().double(), wb) < TOL_OP
        one_by_one = torch.nn.functional.conv2d(x.double(), centre.double()[:, :, None, None], bias[n_split:].double(), stride=2)
        assert rel(b2.cpu().double(), one_by_one) < TOL_OP


@pytest.mark.parametrize("case", ["huge_value", "tiny", "zeros", "band_gaussian_3e-3", "nan"])
def test_stride_two_value_ranges(case):
    lib = hip_engine.load_library()
    B, H, W, Cin, Cout = 6, 8, 8, 128, 64
    g = torch.Generator().manual_seed(47)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    if case == "huge_value":
        x[0, 5, 4, 7] = 3.0e6
    elif case == "tiny":
        x *= 1.0e-6
    elif case == "zeros":
        x[0] = 0.0
        x[1] = 0.0
    elif case == "band_gaussian_3e-3":
        x *= 3.0e-3
    elif case == "nan":
        x[2, 9, 5, 3] = float("nan")     # read by the outputs (2, 2) and (3, 2) x (1, 1) and (2, 1): rows 4..6 -> oy 2, 3; columns 2..4 -> ox 1, 2
    got = conv3x3s2(lib, x.to(DEV), w.to(DEV)).cpu().double()
    if case == "nan":
        hit = torch.zeros(B, H // 2, W // 2, dtype=torch.bool)
        hit[2, 2:4, 1:3] = True
        assert torch.isnan(got.permute(0, 2, 3, 1)[hit]).all()
        x[2, 9, 5, 3] = 0.0
        want = reference_s2(x, w, None, 0, False)
        keep = ~hit[:, None].expand_as(got)
        assert float((got[keep] - want[keep]).abs().max()) <= TOL_OP * float(want.abs().max())
        return
    want = reference_s2(x, w, None, 0, False)
    assert torch.isfinite(got).all()
    for b in range(B):
        scale = max(float(want[b].abs().max()), 1e-30)
        assert float((got[b] - want[b]).abs().max()) <= TOL_OP * scale, (case, b)


def test_stride_two_refuses_what_it_does_not_take():
    lib = hip_engine.load_library()
    with pytest.raises(RuntimeError, match="not a shape of the window-staged kernel at stride 2"):
        conv3x3s2(lib, torch.zeros(1, 64, 32, 32, device=DEV), torch.ones(32, 64, 3, 3, device=DEV))     # a 256-pixel output map
    with pytest.raises(RuntimeError, match="n_split"):
        conv3x3s2(lib, torch.zeros(1, 64, 8, 8, device=DEV), torch.ones(64, 64, 3, 3, device=DEV), n_split=48)
