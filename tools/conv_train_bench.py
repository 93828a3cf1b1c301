"""Times the native 3x3 stride-1 training convolutions (forward, data gradient, weight gradient) against torch's own
(MIOpen) on the layer shapes of the network.  Runs on the GPU box:  python tools/conv_train_bench.py [--size 64] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodelscustom_amd import hip_engine  # noqa: E402
from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--native-only", action="store_true", help="skip torch's convolutions (for a kernel trace of the native ones)")
    a = ap.parse_args()
    s = a.size // 4
    B = a.batch
    layers = [("layer1 (x4)", 64, 64, s, 4), ("layer2 (x3)", 128, 128, s // 2, 3), ("layer3 (x3)", 256, 256, s // 4, 3),
              ("layer4 (x3)", 512, 512, s // 8, 3), ("decoder 0", 512, 256, s // 4, 1), ("decoder 1", 256, 128, s // 2, 1),
              ("decoder 2", 128, 64, s, 1), ("decoder 3", 64, 64, 2 * s, 1)]
    lib = hip_engine.load_library()
    dev = "cuda:0"
    tot_n = tot_t = 0.0
    print(f"{'layer':14s} {'shape':>22s} | native fwd dgrad wgrad (us) | torch fwd  bwd (us)")
    for name, cin, cout, hw, count in layers:
        x = torch.randn(B, cin, hw, hw, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        dy = torch.randn(B, cout, hw, hw, device=dev)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
        ws = torch.empty(max(lib.danra_conv3x3_train_workspace(B, cin, cout, hw, hw) // 4, 1), device=dev)
        args = (B, cin, cout, hw, hw, _ptr(ws), _stream())
        tf = timed(lambda: _check(lib, lib.danra_conv3x3_train_fwd(_ptr(x), _ptr(w), _ptr(y), *args), "f"))
        td = timed(lambda: _check(lib, lib.danra_conv3x3_train_dgrad(_ptr(dy), _ptr(w), _ptr(dx), *args), "d"))
        tw = timed(lambda: _check(lib, lib.danra_conv3x3_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), *args), "w"))
        ttf = ttb = 0.0
        if not a.native_only:
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            ttf = timed(lambda: torch.nn.functional.conv2d(xr, wr, None, 1, 1))
            yy = torch.nn.functional.conv2d(xr, wr, None, 1, 1)
            ttb = timed(lambda: torch.autograd.grad(yy, (xr, wr), dy, retain_graph=True))
        print(f"{name:14s} {str((B, cin, cout, hw, hw)):>22s} | {tf:8.1f} {td:6.1f} {tw:6.1f}      | {ttf:8.1f} {ttb:6.1f}")
        tot_n += count * (tf + td + tw)
        tot_t += count * (ttf + ttb)
    print(f"all 17 layers, three directions: native {tot_n:.0f} us, torch {tot_t:.0f} us")


if __name__ == "__main__":
    main()
