"""Times the elucidated-sampler step kernels of libdanra_hip.so against their HBM byte counts, and one whole sampling
step (everything but the network) against the same step written as the reference's torch statements.

    python tools/edm_bench.py [--batch 32] [--size 256] [--iters 50]

Algorithmic traffic per element (4-byte values): churn 1 read + 2 writes (in-kernel noise); Euler step 2 reads + 4
writes; Heun correction 4 reads + 2 writes; preconditioning 2 + 1; dynamic threshold 1 read; final clamp 1 + 1.
"""
import argparse
import json
import os
import sys
from math import sqrt

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodelscustom_amd import hip_engine  # noqa: E402

PEAK_GBS = 8000.0


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = "cuda:0"
    shape = (args.batch, 3, args.size, args.size)
    n = args.batch * 3 * args.size * args.size
    t = [torch.randn(shape, device=dev) for _ in range(8)]
    images, hat, net_in, net, out, d, nxt, res = t
    thr = torch.empty(args.batch, device=dev)
    sigma, gamma, sigma_next, sd = 3.7, 0.4142135679721832, 2.9, 0.5
    sigma_hat = sigma + gamma * sigma
    k = sqrt(sigma_hat ** 2 - sigma ** 2)
    rows = {}

    def row(name, tensors, fn):
        us = timed(fn, args.iters)
        gbs = tensors * n * 4 / us / 1e3
        rows[name] = {"us": round(us, 2), "algorithmic_GB_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_GBS, 3)}

    row("edm_perturb (in-kernel noise)", 3, lambda: hip_engine.edm_perturb(images, hat, net_in, 1.003, k, 0.27, seed=1, offset=2))
    row("edm_perturb (given noise)", 4, lambda: hip_engine.edm_perturb(images, hat, net_in, 1.003, k, 0.27, noise=net))
    row("edm_precondition", 3, lambda: hip_engine.edm_precondition(hat, net, out, 0.02, 0.49, False))
    row("edm_dynamic_threshold", 1, lambda: hip_engine.edm_dynamic_threshold(out, 0.95, out=thr))
    row("edm_euler (forms the model output)", 6,
        lambda: hip_engine.edm_euler(hat, net, out, None, d, nxt, net_in, 0.02, 0.49, True, sigma_hat, sigma_next - sigma_hat, 0.3))
    row("edm_euler (thresholds a stored model output)", 6,
        lambda: hip_engine.edm_euler(hat, None, out, thr, d, nxt, net_in, 0., 0., False, sigma_hat, sigma_next - sigma_hat, 0.3))
    row("edm_heun", 6, lambda: hip_engine.edm_heun(hat, nxt, net, out, None, d, res, 0.03, 0.49, True, sigma_next,
                                                  0.5 * (sigma_next - sigma_hat)))
    row("edm_finish", 2, lambda: hip_engine.edm_finish(res, True))

    # one step without its two network evaluations: the library's kernels against the reference's statements in torch
    def native_step(dynamic):
        hip_engine.edm_perturb(images, hat, net_in, 1.003, k, 0.27, seed=1, offset=2)
        if dynamic:
            hip_engine.edm_precondition(hat, net, out, 0.02, 0.49, False)
            hip_engine.edm_dynamic_threshold(out, 0.95, out=thr)
        hip_engine.edm_euler(hat, None if dynamic else net, out, thr if dynamic else None, d, nxt, net_in, 0.02, 0.49,
                             not dynamic, sigma_hat, sigma_next - sigma_hat, 0.3)
        if dynamic:
            hip_engine.edm_precondition(nxt, net, res, 0.03, 0.49, False)
            hip_engine.edm_dynamic_threshold(res, 0.95, out=thr)
        hip_engine.edm_heun(hat, nxt, None if dynamic else net, res, thr if dynamic else None, d, images, 0.03, 0.49,
                            not dynamic, sigma_next, 0.5 * (sigma_next - sigma_hat))

    def threshold(x, dynamic):
        if not dynamic:
            return x.clamp(-1., 1.)
        s = torch.quantile(x.flatten(1).abs(), 0.95, dim=-1).clamp_(min=1.).reshape(-1, 1, 1, 1)
        return x.clamp(-s, s) / s

    def torch_step(dynamic):
        nonlocal images
        eps = 1.003 * torch.randn(shape, device=dev)
        added = k * eps
        h = images + added
        sg = torch.full((args.batch,), sigma_hat, device=dev).reshape(-1, 1, 1, 1)
        x_in = (1 * (sg ** 2 + sd ** 2) ** -0.5) * h
        o = threshold((sd ** 2) / (sg ** 2 + sd ** 2) * h + sg * sd * (sd ** 2 + sg ** 2) ** -0.5 * net, dynamic)
        dd = (h - o) / sigma_hat
        nx = h + (sigma_next - sigma_hat) * dd
        sg = torch.full((args.batch,), sigma_next, device=dev).reshape(-1, 1, 1, 1)
        x_in = (1 * (sg ** 2 + sd ** 2) ** -0.5) * nx  # noqa: F841
        o2 = threshold((sd ** 2) / (sg ** 2 + sd ** 2) * nx + sg * sd * (sd ** 2 + sg ** 2) ** -0.5 * net, dynamic)
        dp = (nx - o2) / sigma_next
        images = h + 0.5 * (sigma_next - sigma_hat) * (dd + dp)

    steps = {}
    for dynamic in (False, True):
        key = "dynamic threshold" if dynamic else "static clamp"
        nat = timed(lambda: native_step(dynamic), args.iters)
        ref = timed(lambda: torch_step(dynamic), max(3, args.iters // 10))
        steps[key] = {"native_us": round(nat, 1), "torch_statements_us": round(ref, 1), "ratio": round(ref / nat, 1)}
    print(json.dumps({"workload": f"EDM sampler step arithmetic, batch {args.batch}, 3x{args.size}x{args.size}, fp32",
                      "kernels": rows, "step_without_network": steps}, indent=1))


if __name__ == "__main__":
    main()
