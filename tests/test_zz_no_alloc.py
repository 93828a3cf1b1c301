"""danra_instance_norm_noalloc (the entry point whose scratch is an array built into the library) against
danra_instance_norm (which allocates its scratch per call), through the C ABI: bit-identical results on two calls in a
row, the second on a smaller input (the scratch still holds the first call's statistics), on maps at and above the
1024-pixel slicing limit; oversize calls are refused with a clear message."""
import ctypes

import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from diffusionmodelscustom_amd.hip_engine import _ptr, _stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def both(lib, x):
    B, H, W, C = x.shape
    want, got = torch.empty_like(x), torch.full_like(x, float("nan"))
    assert lib.danra_instance_norm(_ptr(x), _ptr(want), B, H * W, C, _stream()) == 0, lib.danra_last_error()
    assert lib.danra_instance_norm_noalloc(_ptr(x), _ptr(got), B, H * W, C, _stream()) == 0, lib.danra_last_error()
    torch.cuda.synchronize()
    return got, want


def test_no_alloc_instance_norm_equals_the_allocating_entry_point():
    lib = hip_engine.load_library()
    g = torch.Generator().manual_seed(5)
    shapes = [(6, 32, 32, 64),    # first call
              (2, 16, 16, 128),   # second call, smaller: stale statistics of the first call sit in the scratch
              (3, 64, 64, 64),    # 4096 pixels: sliced statistics + merge through the scratch's partial region
              (1, 8, 8, 32),      # smaller again, few blocks: sliced further
              (5, 2, 2, 512)]
    for shape in shapes:
        x = (torch.randn(shape, generator=g) * 3 + 1.5).to(DEV)   # NHWC
        got, want = both(lib, x)
        assert torch.equal(got, want), shape
        ref = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2).double().cpu(), eps=1e-5).permute(0, 2, 3, 1)
        err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, (shape, err)
    # the same call twice gives the same bits
    x = torch.randn((4, 32, 32, 64), generator=g).to(DEV)
    a, _ = both(lib, x)
    b, _ = both(lib, x)
    assert torch.equal(a, b)


def scratch_floats(B, HW, C):
    """The entry point's documented scratch need: statistics, a fixed reserve, and sliced partial statistics for maps
    above 1024 pixels."""
    S = 1
    while HW // S > 1024 or HW % S:
        S *= 2
    return 2 * B * C + 16384 + (2 * B * C * S if S > 1 else 0)


def test_no_alloc_instance_norm_refuses_what_does_not_fit():
    """Refusals happen before any launch, so these calls may name buffers far smaller than their shapes."""
    lib = hip_engine.load_library()
    x = torch.zeros(16, device=DEV)
    for B, HW, C in ((2048, 4, 512),          # statistics alone exceed the array
                     (512, 128 * 128, 64)):   # a 128x128 map: 16 slices of partial statistics per image
        assert scratch_floats(B, HW, C) > (1 << 20) and B * HW < (1 << 24)
        rc = lib.danra_instance_norm_noalloc(_ptr(x), _ptr(x), B, HW, C, _stream())
        assert rc == -1 and b"scratch" in lib.danra_last_error(), (B, HW, C)
    assert lib.danra_instance_norm_noalloc(None, _ptr(x), 1, 4, 32, _stream()) == -1    # null input
    assert lib.danra_instance_norm_noalloc(_ptr(x), _ptr(x), 1, 4, 48, _stream()) == -1  # C not a multiple of 32
    assert lib.danra_instance_norm_noalloc(_ptr(x), _ptr(x), 0, 4, 32, _stream()) == -1  # empty batch


def test_no_alloc_entry_point_allocates_nothing_and_runs_under_graph_capture():
    """What the entry point exists for: the device's free memory does not move across its calls, and it can be captured
    into a graph (an allocating call cannot); the replay equals the allocating entry point bit for bit."""
    lib = hip_engine.load_library()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((4, 64, 64, 64), generator=g).to(DEV)      # sliced statistics: every region of the scratch in use
    y = torch.empty_like(x)
    want = torch.empty_like(x)
    B, H, W, C = x.shape
    assert lib.danra_instance_norm(_ptr(x), _ptr(want), B, H * W, C, _stream()) == 0
    assert lib.danra_instance_norm_noalloc(_ptr(x), _ptr(y), B, H * W, C, _stream()) == 0   # warm call: scratch resolved
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        assert lib.danra_instance_norm_noalloc(_ptr(x), _ptr(y), B, H * W, C, _stream()) == 0
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, f"free device memory moved by {free_before - free_after} bytes"
    assert torch.equal(y, want)

    y.fill_(float("nan"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = lib.danra_instance_norm_noalloc(_ptr(x), _ptr(y), B, H * W, C, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0, lib.danra_last_error()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.isnan(y).all(), "capture must not execute the kernels"
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    x.mul_(0.5).add_(1.0)                                          # new input, same buffers: replay again
    assert lib.danra_instance_norm(_ptr(x), _ptr(want), B, H * W, C, _stream()) == 0
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want)
