"""Times torch's own kernels (MIOpen) for the convolutions of the training step that the native library does not take:
the 8x8 stride-2 stem pair, the 3x3 stride-2 and 1x1 stride-2 convolutions, the transposed convolutions and the final
3x3 convolution.  Runs on the GPU box:  python tools/torch_conv_costs.py [--batch 32]"""
import argparse

import torch


def timed(fn, iters=20):
    for _ in range(5):
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
    ap.add_argument("--batch", type=int, default=32)
    B = ap.parse_args().batch
    dev = "cuda:0"
    layers = [("stem conv1 8x8 s2", torch.nn.Conv2d(4, 64, 8, 2, 3, bias=False), (B, 4, 64, 64)),
              ("stem conv2 8x8 s2", torch.nn.Conv2d(64, 64, 8, 2, 3, bias=False), (B, 64, 32, 32)),
              ("layer2 3x3 s2", torch.nn.Conv2d(64, 128, 3, 2, 1, bias=False), (B, 64, 16, 16)),
              ("layer3 3x3 s2", torch.nn.Conv2d(128, 256, 3, 2, 1, bias=False), (B, 128, 8, 8)),
              ("layer4 3x3 s2", torch.nn.Conv2d(256, 512, 3, 2, 1, bias=False), (B, 256, 4, 4)),
              ("layer2 1x1 s2", torch.nn.Conv2d(64, 128, 1, 2, 0, bias=False), (B, 64, 16, 16)),
              ("layer3 1x1 s2", torch.nn.Conv2d(128, 256, 1, 2, 0, bias=False), (B, 128, 8, 8)),
              ("layer4 1x1 s2", torch.nn.Conv2d(256, 512, 1, 2, 0, bias=False), (B, 256, 4, 4)),
              ("dec0 convT", torch.nn.ConvTranspose2d(512, 512, 2, 2), (B, 512, 2, 2)),
              ("dec1 convT", torch.nn.ConvTranspose2d(256, 256, 2, 2), (B, 256, 4, 4)),
              ("dec2 convT", torch.nn.ConvTranspose2d(128, 128, 2, 2), (B, 128, 8, 8)),
              ("dec3 convT", torch.nn.ConvTranspose2d(64, 64, 2, 2), (B, 64, 16, 16)),
              ("final convT", torch.nn.ConvTranspose2d(64, 64, 2, 2), (B, 64, 32, 32)),
              ("final conv 64->1", torch.nn.Conv2d(64, 1, 3, 1, 1), (B, 64, 64, 64))]
    tot = 0.0
    for name, mod, shape in layers:
        mod = mod.to(dev)
        x = torch.randn(*shape, device=dev, requires_grad=True)
        y = mod(x)
        dy = torch.randn_like(y)
        params = [x] + list(mod.parameters())
        tf = timed(lambda: mod(x))
        tb = timed(lambda: torch.autograd.grad(y, params, dy, retain_graph=True))
        tot += tf + tb
        print(f"{name:18s} {str(shape):>18s}  fwd {tf:6.1f} us  bwd {tb:6.1f} us")
    print(f"total {tot:.0f} us")


if __name__ == "__main__":
    main()
