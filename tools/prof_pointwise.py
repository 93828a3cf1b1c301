"""Calls the pointwise training kernels (1x1 stride-2, transposed 2x2) on the layer shapes of the 64x64 network, for a kernel
trace:  rocprofv3 --kernel-trace ... -- python3 tools/prof_pointwise.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodelscustom_amd import hip_engine  # noqa: E402
from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream  # noqa: E402

lib = hip_engine.load_library()
dev = "cuda:0"
B = 32
layers = [(0, 64, 128, 16), (0, 128, 256, 8), (0, 256, 512, 4), (1, 512, 512, 2), (1, 256, 256, 4), (1, 128, 128, 8),
          (1, 64, 64, 16), (1, 64, 64, 32)]
for kind, cin, cout, hw in layers:
    x = torch.randn(B, cin, hw, hw, device=dev)
    if kind == 0:
        w = torch.randn(cout, cin, 1, 1, device=dev)
        y = torch.empty(B, cout, hw // 2, hw // 2, device=dev)
    else:
        w = torch.randn(cin, cout, 2, 2, device=dev)
        y = torch.empty(B, cout, 2 * hw, 2 * hw, device=dev)
    dy, dx, dw = torch.randn_like(y), torch.empty_like(x), torch.empty_like(w)
    ws = torch.empty(max(lib.danra_pointwise_train_workspace(kind, B, cin, cout, hw, hw) // 4, 1), device=dev)
    for _ in range(5):
        _check(lib, lib.danra_pointwise_train_fwd(kind, _ptr(x), _ptr(w), None, _ptr(y), B, cin, cout, hw, hw, _stream()), "f")
        _check(lib, lib.danra_pointwise_train_dgrad(kind, _ptr(dy), _ptr(w), _ptr(dx), B, cin, cout, hw, hw, _ptr(ws), _stream()), "d")
        _check(lib, lib.danra_pointwise_train_wgrad(kind, _ptr(x), _ptr(dy), _ptr(dw), B, cin, cout, hw, hw, _ptr(ws), _stream()), "w")
    torch.cuda.synchronize()
