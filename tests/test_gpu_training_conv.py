"""The training step's 3x3 stride-1 convolutions in the native library (danra_conv3x3_train_fwd / _dgrad / _wgrad)
against torch's own arithmetic.

The checker is what the reference's training loop runs for these layers - torch.nn.functional.conv2d and its autograd
backward (torchvision BasicBlock conv1 / conv2 and the decoder blocks' nn.Conv2d, modules_DANRA_conditional.py:154,417;
training_DANRA_conditional.py:962-984) - evaluated in float64.  Tolerance: 1e-5 * max|.| per tensor, the per-op bar of
the inference kernels.
"""
import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_native(x, w, dy):
    lib = hip_engine.load_library()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert lib.danra_conv3x3_train_supported(B, Cin, Cout, H, W)
    ws = torch.empty(max(lib.danra_conv3x3_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=DEV)
    y = torch.empty((B, Cout, H, W), device=DEV)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    _check(lib, lib.danra_conv3x3_train_fwd(_ptr(x), _ptr(w), _ptr(y), B, Cin, Cout, H, W, _ptr(ws), _stream()), "fwd")
    _check(lib, lib.danra_conv3x3_train_dgrad(_ptr(dy), _ptr(w), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws), _stream()), "dgrad")
    _check(lib, lib.danra_conv3x3_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws), _stream()), "wgrad")
    torch.cuda.synchronize()
    return y, dx, dw


def close(got, want, what, tol=1e-5):
    err = (got.double() - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= tol * ref, f"{what}: max error {err:.3e} against max |.| {ref:.3e} (tolerance {tol:g} relative)"


# every 3x3 stride-1 convolution shape of the 64x64 and 128x128 networks (batch sizes chosen so that tiles are cut:
# 5 images of 2x2 leave an eleven-image hole in the only pixel tile, 3 images of 4x4 one of one image), a batch large
# enough for the pixel slices of the weight gradient, and a rectangular map
SHAPES = [
    (3, 64, 64, 16, 16), (2, 128, 128, 8, 8), (3, 256, 256, 4, 4), (5, 512, 512, 2, 2),
    (3, 512, 256, 4, 4), (2, 256, 128, 8, 8), (2, 128, 64, 16, 16), (2, 64, 64, 32, 32),
    (1, 64, 64, 64, 64), (32, 64, 64, 16, 16), (32, 512, 512, 2, 2), (2, 64, 128, 8, 16),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W", SHAPES)
def test_convolution_kernels_match_float64(B, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(B * 1000 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(DEV)
    dy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    y, dx, dw = run_native(x, w, dy)
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)  # the float64 arbiter, on the CPU
    want = torch.nn.functional.conv2d(xd, wd, None, 1, 1)
    want.backward(dy.double().cpu())
    close(y.cpu(), want.detach(), "forward")
    close(dx.cpu(), xd.grad, "data gradient")
    close(dw.cpu(), wd.grad, "weight gradient")
    again = run_native(x, w, dy)
    for a, b, what in zip((y, dx, dw), again, ("forward", "data gradient", "weight gradient")):
        assert torch.equal(a, b), f"{what}: two calls differ"


def test_unsupported_shapes_are_refused():
    lib = hip_engine.load_library()
    for B, Cin, Cout, H, W in [(2, 64, 1, 64, 64), (2, 4, 64, 64, 64), (2, 64, 64, 128, 128), (2, 64, 64, 6, 6),
                               (2, 64, 64, 3, 16), (0, 64, 64, 8, 8)]:
        assert not lib.danra_conv3x3_train_supported(B, Cin, Cout, H, W), (B, Cin, Cout, H, W)
        assert lib.danra_conv3x3_train_workspace(B, Cin, Cout, H, W) == 0
    x = torch.zeros(2, 64, 6, 6, device=DEV)
    w = torch.zeros(64, 64, 3, 3, device=DEV)
    rc = lib.danra_conv3x3_train_fwd(_ptr(x), _ptr(w), _ptr(x), 2, 64, 64, 6, 6, _ptr(x), _stream())
    assert rc != 0 and b"danra_conv3x3_train_supported" in lib.danra_last_error()


def test_conv2d_function_equals_the_torch_module(monkeypatch):
    """hip_engine.conv2d on an nn.Conv2d with a bias (a decoder block's convolution): output and all three gradients
    against the module itself, and the module is what runs when the shape or the mode is not the library's."""
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(128, 64, 3, 1, 1).to(DEV)
    x = torch.randn(4, 128, 16, 16, device=DEV, requires_grad=True)
    dy = torch.randn(4, 64, 16, 16, device=DEV)
    calls = []
    lib = hip_engine.load_library()
    for name in ("danra_conv3x3_train_fwd", "danra_conv3x3_train_dgrad", "danra_conv3x3_train_wgrad"):
        real = getattr(lib, name)
        monkeypatch.setattr(lib, name, lambda *a, _r=real, _n=name: (calls.append(_n), _r(*a))[1])
    y = hip_engine.conv2d(conv, x)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    assert sorted(calls) == ["danra_conv3x3_train_dgrad", "danra_conv3x3_train_fwd", "danra_conv3x3_train_wgrad"]
    conv64 = torch.nn.Conv2d(128, 64, 3, 1, 1).double()
    conv64.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = conv64(x64)
    y64.backward(dy.double().cpu())
    for a, b, what in zip(got, (y64.detach(), x64.grad, conv64.weight.grad, conv64.bias.grad),
                          ("output", "input gradient", "weight gradient", "bias gradient")):
        close(a.cpu(), b, what)
    del calls[:]
    with torch.no_grad():
        hip_engine.conv2d(conv, x)                                    # no gradient: the module
    hip_engine.conv2d(torch.nn.Conv2d(128, 64, 3, 2, 0).to(DEV), x)   # no padding: the module
    hip_engine.conv2d(torch.nn.Conv2d(128, 32, 3, 1, 1).to(DEV), x)   # 32 output channels: the module
    assert calls == []
    hip_engine.conv2d(torch.nn.Conv2d(128, 64, 3, 2, 1).to(DEV), x)   # stride 2, padding 1: kernels of their own
    assert calls == []


# the 1x1 stride-2 shortcut convolutions (kind 0) and the kernel-2 stride-2 transposed convolutions (kind 1) of the 64x64
# network, odd batch sizes (the last 64-pixel chunk is cut), and the final block's transposed convolution at batch 8
POINTWISE = [
    (0, 3, 64, 128, 16, 16), (0, 5, 128, 256, 8, 8), (0, 7, 256, 512, 4, 4), (0, 32, 64, 128, 16, 16),
    (1, 5, 512, 512, 2, 2), (1, 3, 256, 256, 4, 4), (1, 3, 128, 128, 8, 8), (1, 2, 64, 64, 16, 16), (1, 8, 64, 64, 32, 32),
    (1, 32, 512, 512, 2, 2), (1, 2, 64, 16, 8, 4),
]


@pytest.mark.parametrize("kind,B,Cin,Cout,H,W", POINTWISE)
def test_pointwise_kernels_match_float64(kind, B, Cin, Cout, H, W):
    lib = hip_engine.load_library()
    assert lib.danra_pointwise_train_supported(kind, B, Cin, Cout, H, W)
    g = torch.Generator().manual_seed(kind * 7 + B * 1000 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    if kind == 0:
        w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
        bias = None
        ref = lambda xx, ww, bb: torch.nn.functional.conv2d(xx, ww, None, 2, 0)
    else:
        w = torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5
        bias = torch.randn(Cout, generator=g)
        ref = lambda xx, ww, bb: torch.nn.functional.conv_transpose2d(xx, ww, bb, 2, 0)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want = ref(xd, wd, None if bias is None else bias.double())
    dy = torch.randn(want.shape, generator=g)
    want.backward(dy.double())

    def run():
        xg, wg, dyg = x.to(DEV), w.to(DEV), dy.to(DEV)
        bg = None if bias is None else bias.to(DEV)
        y = torch.empty(want.shape, device=DEV)
        dx, dw = torch.full_like(xg, float("nan")), torch.empty_like(wg)
        ws = torch.empty(max(lib.danra_pointwise_train_workspace(kind, B, Cin, Cout, H, W) // 4, 1), device=DEV)
        _check(lib, lib.danra_pointwise_train_fwd(kind, _ptr(xg), _ptr(wg), _ptr(bg), _ptr(y), B, Cin, Cout, H, W, _stream()), "fwd")
        _check(lib, lib.danra_pointwise_train_dgrad(kind, _ptr(dyg), _ptr(wg), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws), _stream()),
               "dgrad")
        _check(lib, lib.danra_pointwise_train_wgrad(kind, _ptr(xg), _ptr(dyg), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws), _stream()),
               "wgrad")
        torch.cuda.synchronize()
        return y.cpu(), dx.cpu(), dw.cpu()

    y, dx, dw = run()
    close(y, want.detach(), "forward")
    close(dx, xd.grad, "data gradient")
    close(dw, wd.grad, "weight gradient")
    for a, b, what in zip((y, dx, dw), run(), ("forward", "data gradient", "weight gradient")):
        assert torch.equal(a, b), f"{what}: two calls differ"


def test_pointwise_modules_equal_torch(monkeypatch):
    """hip_engine.conv2d on a 1x1 stride-2 nn.Conv2d and hip_engine.conv_transpose2d on a decoder block's nn.ConvTranspose2d
    (with its bias): outputs and every gradient against the modules in float64; other modules are left to torch."""
    torch.manual_seed(4)
    lib = hip_engine.load_library()
    calls = []
    for name in ("danra_pointwise_train_fwd", "danra_pointwise_train_dgrad", "danra_pointwise_train_wgrad"):
        real = getattr(lib, name)
        monkeypatch.setattr(lib, name, lambda *a, _r=real, _n=name: (calls.append(_n), _r(*a))[1])
    for mod, fn, shape in ((torch.nn.Conv2d(64, 128, 1, 2, bias=False), hip_engine.conv2d, (4, 64, 16, 16)),
                           (torch.nn.ConvTranspose2d(128, 128, 2, 2), hip_engine.conv_transpose2d, (4, 128, 8, 8))):
        mod = mod.to(DEV)
        x = torch.randn(*shape, device=DEV, requires_grad=True)
        del calls[:]
        y = fn(mod, x)
        dy = torch.randn_like(y)
        grads = torch.autograd.grad(y, [x] + list(mod.parameters()), dy)
        assert sorted(calls) == ["danra_pointwise_train_dgrad", "danra_pointwise_train_fwd", "danra_pointwise_train_wgrad"]
        mod64 = type(mod)(mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, bias=mod.bias is not None).double()
        mod64.load_state_dict({k: v.double().cpu() for k, v in mod.state_dict().items()})
        x64 = x.detach().double().cpu().requires_grad_(True)
        y64 = mod64(x64)
        want = torch.autograd.grad(y64, [x64] + list(mod64.parameters()), dy.double().cpu())
        close(y.detach().cpu(), y64.detach(), "output")
        for a, b in zip(grads, want):
            close(a.cpu(), b, "gradient")
    del calls[:]
    x = torch.randn(2, 64, 16, 16, device=DEV, requires_grad=True)
    hip_engine.conv2d(torch.nn.Conv2d(64, 128, 1, 1, bias=False).to(DEV), x)            # stride 1: the module
    hip_engine.conv_transpose2d(torch.nn.ConvTranspose2d(64, 64, 4, 2, 1).to(DEV), x)   # kernel 4: the module
    with torch.no_grad():
        hip_engine.conv_transpose2d(torch.nn.ConvTranspose2d(64, 64, 2, 2).to(DEV), x)  # no gradient: the module
    assert calls == []


# the three stride-2 convolutions of the 64x64 and 128x128 networks (whole-image tiles, 4 and 16 images per tile with batches
# that leave a tile part empty, row tiles), a rectangular map, a batch large enough for the slabs of every sum, other channel
# counts
STRIDE2_SHAPES = [
    (3, 64, 128, 16, 16), (2, 128, 256, 8, 8), (5, 256, 512, 4, 4), (32, 64, 128, 16, 16), (32, 128, 256, 8, 8),
    (32, 256, 512, 4, 4), (2, 64, 128, 32, 32), (1, 64, 128, 64, 64), (3, 32, 64, 8, 16), (2, 96, 192, 16, 8), (17, 64, 64, 4, 4),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W", STRIDE2_SHAPES)
def test_stride_two_convolution_kernels_match_float64(B, Cin, Cout, H, W):
    """nn.Conv2d(Cin, Cout, 3, stride 2, padding 1, bias=False) - conv1 of the first BasicBlock of layer2 / 3 / 4 (torchvision
    BasicBlock under modules_DANRA_conditional.py:154) - in the library: forward, data gradient and weight gradient against
    float64 conv2d on the CPU, bit-identical between calls; then through the module hook."""
    lib = hip_engine.load_library()
    assert lib.danra_conv3x3s2_train_supported(B, Cin, Cout, H, W)
    g = torch.Generator().manual_seed(B * 1000 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(DEV)
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g).to(DEV)
    ws = torch.empty(max(lib.danra_conv3x3s2_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=DEV)

    def run():
        y, dx, dw = torch.full_like(dy, float("nan")), torch.full_like(x, float("nan")), torch.full_like(w, float("nan"))
        _check(lib, lib.danra_conv3x3s2_train_fwd(_ptr(x), _ptr(w), _ptr(y), B, Cin, Cout, H, W, _ptr(ws), _stream()), "fwd")
        _check(lib, lib.danra_conv3x3s2_train_dgrad(_ptr(dy), _ptr(w), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws), _stream()), "dgrad")
        _check(lib, lib.danra_conv3x3s2_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws), _stream()), "wgrad")
        torch.cuda.synchronize()
        return y, dx, dw
    y, dx, dw = run()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    want = torch.nn.functional.conv2d(xd, wd, None, 2, 1)
    want.backward(dy.double().cpu())
    close(y.cpu(), want.detach(), "forward")
    close(dx.cpu(), xd.grad, "data gradient")
    close(dw.cpu(), wd.grad, "weight gradient")
    for a, b, what in zip((y, dx, dw), run(), ("forward", "data gradient", "weight gradient")):
        assert torch.equal(a, b), f"{what}: two calls differ"
    conv = torch.nn.Conv2d(Cin, Cout, 3, stride=2, padding=1, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
    xr = x.clone().requires_grad_(True)
    calls = []
    real = hip_engine._Conv3x3S2Train.apply
    try:
        hip_engine._Conv3x3S2Train.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        out = hip_engine.conv2d(conv, xr)
    finally:
        hip_engine._Conv3x3S2Train.apply = real
    assert calls, "the native kernels did not run"
    out.backward(dy)
    assert torch.equal(out.detach(), y) and torch.equal(xr.grad, dx) and torch.equal(conv.weight.grad, dw)


def test_stride_two_convolution_refusals_and_fallback():
    """Shapes the stride-2 kernels refuse; hip_engine.conv2d then evaluates the convolution some other way (the stride-1
    kernels on the whole map where they take it, else the torch module) and the result is still the convolution."""
    lib = hip_engine.load_library()
    for shape in [(2, 64, 128, 24, 24), (2, 48, 128, 16, 16), (2, 64, 96, 16, 16), (2, 64, 128, 15, 16), (2, 64, 128, 2, 2),
                  (0, 64, 128, 16, 16)]:
        assert not lib.danra_conv3x3s2_train_supported(*shape), shape
        assert lib.danra_conv3x3s2_train_workspace(*shape) == 0
    x0 = torch.zeros(2, 64, 24, 24, device=DEV)
    w0 = torch.zeros(128, 64, 3, 3, device=DEV)
    rc = lib.danra_conv3x3s2_train_fwd(_ptr(x0), _ptr(w0), _ptr(x0), 2, 64, 128, 24, 24, _ptr(x0), _stream())
    assert rc != 0 and b"danra_conv3x3s2_train_supported" in lib.danra_last_error()
    B, Cin, Cout, H, W = 2, 64, 128, 24, 24   # the 12x12 level of a 96x96 input
    g = torch.Generator().manual_seed(5)
    conv = torch.nn.Conv2d(Cin, Cout, 3, stride=2, padding=1, bias=False).to(DEV)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g).to(DEV)
    y = hip_engine.conv2d(conv, x)
    y.backward(dy)
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = conv.weight.detach().double().cpu().requires_grad_(True)
    want = torch.nn.functional.conv2d(xd, wd, None, 2, 1)
    want.backward(dy.double().cpu())
    close(y.detach().cpu(), want.detach(), "forward", 1e-4)
    close(x.grad.cpu(), xd.grad, "data gradient", 1e-4)
    close(conv.weight.grad.cpu(), wd.grad, "weight gradient", 1e-4)


@pytest.mark.parametrize("B,Cin,H,W", [(3, 64, 64, 64), (2, 64, 128, 128), (5, 64, 16, 16), (32, 64, 64, 64), (2, 32, 7, 13), (1, 3, 1, 1)])
def test_one_output_channel_convolution_matches_float64(B, Cin, H, W):
    """The network's last convolution (nn.Conv2d(64, 1, 3, padding=1), modules_DANRA_conditional.py:415-416 with
    last_layer) in the library, forward, data gradient and weight gradient, against float64 conv2d; then through the
    module hook with the bias."""
    lib = hip_engine.load_library()
    assert lib.danra_conv3x3_c1_train_supported(B, Cin, H, W)
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(1, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(DEV)
    dy = torch.randn(B, 1, H, W, generator=g).to(DEV)
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
    ws = torch.empty(max(lib.danra_conv3x3_c1_train_workspace(B, Cin, H, W) // 4, 1), device=DEV)
    _check(lib, lib.danra_conv3x3_c1_train_fwd(_ptr(x), _ptr(w), _ptr(y), B, Cin, H, W, _stream()), "fwd")
    _check(lib, lib.danra_conv3x3_c1_train_dgrad(_ptr(dy), _ptr(w), _ptr(dx), B, Cin, H, W, _stream()), "dgrad")
    _check(lib, lib.danra_conv3x3_c1_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, H, W, _ptr(ws), _stream()), "wgrad")
    torch.cuda.synchronize()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    want = torch.nn.functional.conv2d(xd, wd, None, 1, 1)
    want.backward(dy.double().cpu())
    close(y.cpu(), want.detach(), "forward")
    close(dx.cpu(), xd.grad, "data gradient")
    close(dw.cpu(), wd.grad, "weight gradient")
    dw2 = torch.empty_like(w)
    _check(lib, lib.danra_conv3x3_c1_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw2), B, Cin, H, W, _ptr(ws), _stream()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "weight gradient: two calls differ"
    # the module hook
    conv = torch.nn.Conv2d(Cin, 1, 3, 1, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
    xr = x.clone().requires_grad_(True)
    calls = []
    real = hip_engine._Conv3x3C1Train.apply
    try:
        hip_engine._Conv3x3C1Train.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        out = hip_engine.conv2d(conv, xr)
    finally:
        hip_engine._Conv3x3C1Train.apply = real
    assert calls, "the native kernels did not run"
    out.backward(dy)
    close(out.detach().cpu(), want.detach() + conv.bias.detach().double().cpu().view(1, 1, 1, 1), "module output")
    close(xr.grad.cpu(), xd.grad, "module data gradient")
    close(conv.weight.grad.cpu(), wd.grad, "module weight gradient")
    close(conv.bias.grad.cpu(), dy.double().cpu().sum().view(1), "module bias gradient")


# the stem convolutions of the 64x64 and 128x128 networks at every count of input planes (x alone .. x + lsm + topo + cond),
# conv2 at both sizes, 32x32 and 16x16 inputs (eight-row tiles), a rectangular map, more channels than the network has,
# and a batch large enough for 256 slabs of the weight gradient
STEM_SHAPES = [
    (3, 4, 64, 64, 64), (2, 3, 64, 64, 64), (2, 2, 64, 64, 64), (2, 1, 64, 64, 64), (2, 64, 64, 32, 32),
    (1, 4, 64, 128, 128), (2, 64, 64, 64, 64), (1, 1, 64, 32, 32), (2, 32, 64, 16, 16), (3, 64, 128, 32, 64),
    (32, 64, 64, 32, 32), (32, 4, 64, 64, 64), (1, 96, 192, 16, 32),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W", STEM_SHAPES)
def test_stem_convolution_kernels_match_float64(B, Cin, Cout, H, W):
    """nn.Conv2d(Cin, Cout, 8, stride 2, padding 3, bias=False) - the encoder's conv1 and conv2
    (modules_DANRA_conditional.py:186-191) - in the library: forward, weight gradient and (32-channel multiples) data
    gradient against float64 conv2d on the CPU, bit-identical between calls; then through the module hook."""
    lib = hip_engine.load_library()
    assert lib.danra_conv8x8s2_train_supported(B, Cin, Cout, H, W)
    has_dgrad = bool(lib.danra_conv8x8s2_train_dgrad_supported(B, Cin, Cout, H, W))
    assert has_dgrad == (Cin % 32 == 0)
    g = torch.Generator().manual_seed(B * 1000 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 8, 8, generator=g) / (8.0 * Cin ** 0.5)).to(DEV)
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g).to(DEV)
    ws = torch.empty(max(lib.danra_conv8x8s2_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=DEV)

    def run():
        y, dx, dw = torch.empty_like(dy), torch.full_like(x, float("nan")), torch.empty_like(w)
        _check(lib, lib.danra_conv8x8s2_train_fwd(_ptr(x), _ptr(w), _ptr(y), B, Cin, Cout, H, W, _stream()), "fwd")
        if has_dgrad:
            _check(lib, lib.danra_conv8x8s2_train_dgrad(_ptr(dy), _ptr(w), _ptr(dx), B, Cin, Cout, H, W, _stream()), "dgrad")
        _check(lib, lib.danra_conv8x8s2_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws), _stream()), "wgrad")
        torch.cuda.synchronize()
        return y, dx, dw
    y, dx, dw = run()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    want = torch.nn.functional.conv2d(xd, wd, None, 2, 3)
    want.backward(dy.double().cpu())
    close(y.cpu(), want.detach(), "forward")
    if has_dgrad:
        close(dx.cpu(), xd.grad, "data gradient")
    close(dw.cpu(), wd.grad, "weight gradient")
    for a, b, what in zip((y, dx, dw), run(), ("forward", "data gradient", "weight gradient")):
        assert torch.equal(a, b, ) or (what == "data gradient" and not has_dgrad), f"{what}: two calls differ"
    # the module hook: the input asks for a gradient only where the data-gradient kernel exists
    conv = torch.nn.Conv2d(Cin, Cout, 8, 2, 3, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
    xr = x.clone().requires_grad_(has_dgrad)
    calls = []
    real = hip_engine._Conv8x8S2Train.apply
    try:
        hip_engine._Conv8x8S2Train.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        out = hip_engine.conv2d(conv, xr)
    finally:
        hip_engine._Conv8x8S2Train.apply = real
    assert calls, "the native kernels did not run"
    out.backward(dy)
    assert torch.equal(out.detach(), y) and torch.equal(conv.weight.grad, dw)
    if has_dgrad:
        assert torch.equal(xr.grad, dx)


def test_stem_convolution_refusals():
    lib = hip_engine.load_library()
    for B, Cin, Cout, H, W in [(2, 5, 64, 64, 64), (2, 64, 32, 32, 32), (2, 64, 64, 31, 32), (2, 64, 64, 32, 24),
                               (2, 64, 64, 8, 8), (2, 64, 64, 256, 256), (0, 64, 64, 32, 32), (2, 64, 64, 20, 32)]:
        assert not lib.danra_conv8x8s2_train_supported(B, Cin, Cout, H, W), (B, Cin, Cout, H, W)
        assert lib.danra_conv8x8s2_train_workspace(B, Cin, Cout, H, W) == 0
    assert lib.danra_conv8x8s2_train_supported(2, 4, 64, 64, 64) and not lib.danra_conv8x8s2_train_dgrad_supported(2, 4, 64, 64, 64)
    x = torch.zeros(2, 4, 64, 64, device=DEV)
    w = torch.zeros(64, 4, 8, 8, device=DEV)
    dy = torch.zeros(2, 64, 32, 32, device=DEV)
    rc = lib.danra_conv8x8s2_train_dgrad(_ptr(dy), _ptr(w), _ptr(x), 2, 4, 64, 64, 64, _stream())
    assert rc != 0 and b"danra_conv8x8s2_train_dgrad_supported" in lib.danra_last_error()
    # an input that asks for a gradient where no data-gradient kernel exists: the module itself
    conv = torch.nn.Conv2d(4, 64, 8, 2, 3, bias=False).to(DEV)
    xr = torch.randn(2, 4, 64, 64, device=DEV, requires_grad=True)
    out = hip_engine.conv2d(conv, xr)
    out.sum().backward()
    assert xr.grad is not None and torch.isfinite(xr.grad).all()


@pytest.mark.parametrize("B,C,H,W", [(32, 64, 64, 64), (3, 512, 2, 2), (5, 1, 16, 16), (2, 96, 6, 10)])
def test_channel_sum_matches_float64_and_is_reproducible(B, C, H, W):
    """danra_channel_sum (the bias gradient of the decoder's convolutions, autograd's dy.sum((0, 2, 3))) against float64;
    two calls agree bit for bit; a layout the kernel does not take goes through torch's reduction."""
    g = torch.Generator().manual_seed(C + H)
    t = (torch.randn(B, C, H, W, generator=g) + 0.3).to(DEV)
    got = hip_engine.channel_sum(t)
    want = t.double().cpu().sum((0, 2, 3))
    err = (got.double().cpu() - want).abs().max().item()
    assert err <= 1e-6 * t.double().abs().sum((0, 2, 3)).max().item(), err
    assert torch.equal(got, hip_engine.channel_sum(t))
    odd = t[:, :, :, : W - 1]                      # not contiguous: torch's reduction
    assert torch.allclose(hip_engine.channel_sum(odd).cpu(), odd.cpu().sum((0, 2, 3)), rtol=1e-4, atol=1e-4)
