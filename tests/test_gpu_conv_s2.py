"""The strided window kernel (csrc/kernels_conv_s2.hip: the 8x8 stride-2 padding-3 convolution of Encoder.conv2,
modules_DANRA_conditional.py:186-191, 269-273) through its per-op C-ABI entry, danra_conv8x8s2, against a float64
evaluation of torch.nn.functional.conv2d: one and several tiles per map row, rectangular maps, more than one 64-channel
block on either side, the epilogue terms, and the value ranges in which a workgroup must leave the split-operand fp16
products for fp32 arithmetic.  Per-op tolerance: 1e-5 of the result's scale."""
import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from gpu_util import conv8x8s2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_OP = 1e-5


def reference(x, w, bias=None, res=None, relu=False):
    y = torch.nn.functional.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride=2, padding=3)
    if res is not None:
        y = y + res.double()
    return torch.relu(y) if relu else y


def run(x, w, bias=None, res=None, relu=False):
    lib = hip_engine.load_library()
    g = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    return conv8x8s2(lib, g(x), g(w), g(bias), g(res), relu).cpu().double()


def rel(got, want):
    return float((got - want).abs().max() / max(float(want.abs().max()), 1e-30))


SHAPES = [
    (64, 32, 32, 64, 64),     # the network's layer at a 64x64 input: one tile per four map rows
    (4, 64, 64, 64, 64),      # two tiles side by side: windows that start inside the map
    (2, 128, 128, 64, 64),    # interior tiles on all sides
    (3, 16, 32, 64, 128),     # a rectangular map, two channel tiles
    (5, 8, 32, 64, 64),       # one tile per image
    (2, 32, 32, 128, 64),     # two input channel blocks: the window is staged twice
    (1, 32, 64, 192, 128),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_every_shape_against_float64(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H * 100 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 8, 8, generator=g) * (2.0 / (64 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    got = run(x, w, bias, res, relu=True)
    want = reference(x, w, bias, res, relu=True)
    assert got.shape == want.shape
    assert rel(got, want) < TOL_OP
    plain = run(x, w)
    assert rel(plain, reference(x, w)) < TOL_OP
    assert torch.equal(run(x, w, bias, res, relu=True), got)    # bitwise reproducible


def test_every_tap_and_channel_lands_where_conv2d_puts_it():
    """A one-hot input pixel and channel against a weight tensor of distinct integers: every output is one weight, exact
    in any arithmetic, so a wrong tap, parity plane or channel slice shows as a wrong integer."""
    B, H, W, Cin, Cout = 1, 16, 32, 64, 64
    w = (torch.arange(Cout * Cin * 64, dtype=torch.float32).reshape(Cout, Cin, 8, 8) % 2039) - 1000.0
    for (iy, ix, c) in [(0, 0, 0), (5, 9, 17), (15, 31, 63), (8, 16, 32), (3, 30, 5)]:
        x = torch.zeros(B, Cin, H, W)
        x[0, c, iy, ix] = 1.0
        assert torch.equal(run(x, w), reference(x, w)), (iy, ix, c)


def test_unsupported_shapes_are_refused():
    lib = hip_engine.load_library()
    for shape, wshape in (((1, 64, 16, 16), (64, 64, 8, 8)),       # a 16-wide map: narrower than a tile's window
                          ((1, 64, 32, 48), (64, 64, 8, 8)),       # not a power of two
                          ((1, 32, 32, 32), (64, 32, 8, 8)),       # Cin not a multiple of 64
                          ((1, 64, 32, 32), (32, 64, 8, 8))):      # Cout not a multiple of 64
        with pytest.raises(RuntimeError, match="not a shape of the strided window kernel"):
            conv8x8s2(lib, torch.zeros(shape, device=DEV), torch.ones(wshape, device=DEV))


CASES = ["huge_value", "huge_region", "tiny", "zeros", "band_gaussian_3e-3", "band_uniform_2^-8_2^-3", "band_mixed_with_order_one"]


@pytest.mark.parametrize("case", CASES)
def test_value_ranges(case):
    """A workgroup whose staged window holds a value of 2^15 or more, or nothing of 2^-8 or more (unless it is all zero),
    forms its tile with fp32 multiply-adds; everything else multiplies in the split-operand form, including the band
    below 2^-3 where the low halves are fp16 subnormals.  Judged per image."""
    B, H, W, Cin, Cout = 6, 32, 32, 64, 64
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 8, 8, generator=g) * (2.0 / (64 * Cin)) ** 0.5
    if case == "huge_value":
        x[0, 5, H // 2, W - 1] = 3.0e6
    elif case == "huge_region":
        x[1, :, H // 2:, :] *= 1.0e5
    elif case == "tiny":
        x *= 1.0e-6
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
    B, H, W, Cin, Cout = 2, 32, 32, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 8, 8, generator=g) * 0.02
    clean = run(x, w)
    x[1, 7, 13, 20] = float("nan")
    got = run(x, w)
    # output (oy, ox) reads input rows 2 oy - 3 .. 2 oy + 4: row 13 for oy = 5..8, column 20 for ox = 8..11
    hit = torch.zeros(B, H // 2, W // 2, dtype=torch.bool)
    hit[1, 5:9, 8:12] = True
    assert torch.isnan(got.permute(0, 2, 3, 1)[hit]).all()
    keep = ~hit[:, None].expand_as(got)
    assert torch.equal(got[keep], clean[keep])
