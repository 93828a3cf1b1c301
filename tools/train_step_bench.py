"""Training-step timing on one GPU: forward + MSE loss + backward + AdamW step of the conditional DANRA network in train
mode, with the attention cores, train-mode BatchNorm and InstanceNorm in libdanra_hip.so against torch's own operators on the
same device, and the attention core alone (forward, backward) against its algorithmic FLOPs.

    python tools/train_step_bench.py [--size 64] [--batch 32] [--steps 10]

Prints one JSON line.  Not the headline benchmark (that is bench.py: sampled images/s).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusionmodelscustom_amd import DiffusionUtils, hip_engine  # noqa: E402
from diffusionmodelscustom_amd.modules_DANRA_conditional import Decoder, DiffusionNet, Encoder  # noqa: E402
from diffusionmodelscustom_amd.synthetic import fill_state_dict_, synthetic_inputs  # noqa: E402


def build(size, dev):
    lsm, topo = torch.zeros(1, size, size), torch.zeros(1, size, size)
    enc = Encoder(1, 256, lsm_tensor=lsm, topo_tensor=topo, cond_on_img=True, cond_img_dim=(1, size, size),
                  block_layers=[2, 2, 2, 2], num_classes=4, n_heads=4)
    model = DiffusionNet(enc, Decoder(512, 1, 256, 64, n_heads=4))
    fill_state_dict_(model, 42)
    return model.to(dev).train()


def time_steps(model, data, du, steps, warmup):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    x, y, planes = data
    torch.cuda.reset_peak_memory_stats()

    def one():
        t = du.sampleTimesteps(x.shape[0])
        noisy, noise = du.noiseImage(x, t)
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(noisy, t, y, *planes), noise)
        loss.backward()
        opt.step()
        return loss.detach()

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30, float(loss)


def time_core(B, heads, L, iters=20):
    lib = hip_engine.load_library()
    _p, _s, _c = hip_engine._ptr, hip_engine._stream, hip_engine._check
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v = (torch.randn(B * heads, L, 16, device="cuda", generator=g) for _ in range(3))
    dout = torch.randn(B, L, heads * 16, device="cuda", generator=g)
    out, lse = torch.empty_like(dout), torch.empty(B * heads, L, device="cuda")
    dq, dk, dv, delta = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q), torch.empty_like(lse)

    def fwd():
        _c(lib, lib.danra_attention_core_fwd(_p(q), _p(k), _p(v), _p(out), _p(lse), B, heads, L, 16, _s()), "fwd")

    def bwd():
        _c(lib, lib.danra_attention_core_bwd(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(dq), _p(dk), _p(dv),
                                             _p(delta), B, heads, L, 16, _s()), "bwd")

    res = {}
    for name, fn, gemms in (("fwd", fwd, 2), ("bwd", bwd, 5)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        flops = gemms * 2.0 * B * heads * L * L * 16   # algorithmic: 2 matrix products forward, 5 backward
        res[name] = {"us": round(us, 1), "algorithmic_tflops": round(flops / us / 1e6, 1)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=8)  # (solver searches and lazy initialisations take several steps to settle)
    ap.add_argument("--channels-last", action="store_true", help="parameters and activations in torch.channels_last")
    a = ap.parse_args()
    if a.channels_last:
        global build
        plain_build = build

        def build(size, dev):
            return plain_build(size, dev).to(memory_format=torch.channels_last)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    d = synthetic_inputs(a.batch, a.size, a.size, seed=42, conditional=True)
    data = (d["x"].to(dev), d["y"].to(dev), [d["cond_img"].to(dev), d["lsm_cond"].to(dev), d["topo_cond"].to(dev)])
    du = DiffusionUtils(1000, 1e-4, 0.02, device=dev)
    out = {"workload": f"DANRA conditional {a.size}x{a.size} training step (forward + MSE + backward + AdamW), batch {a.batch}, fp32"
                       + (", channels_last" if a.channels_last else ""),
           "steps": a.steps}
    # the first pass through a fresh process is not representative (the first eighteen or so steps run up to 50 % slower than
    # the same steps later on: solver searches, lazy initialisations, code object loads): one discarded pass
    time_steps(build(a.size, dev), data, du, a.steps, a.warmup)
    model = build(a.size, dev)
    ms, gib, loss = time_steps(model, data, du, a.steps, a.warmup)
    out["native_kernels"] = {"ms_per_step": round(ms, 2), "images_per_s": round(a.batch / ms * 1e3, 1),
                                    "peak_GiB": round(gib, 2), "last_loss": loss}
    # the same step replayed from one hipGraph (diffusionmodelscustom_amd.training.GraphedTrainStep)
    from diffusionmodelscustom_amd.training import GraphedTrainStep
    x, y, planes = data

    def graphed():
        model = build(a.size, dev)
        # fused: one multi-tensor kernel per step; the plain capturable form launches ~300 scalar copy kernels per step
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, fused=True)
        step = GraphedTrainStep(model, opt, du, torch.nn.MSELoss(), dict(images=x, seasons=y, cond_images=planes[0],
                                                                         lsm=planes[1], topo=planes[2]))
        torch.cuda.reset_peak_memory_stats()
        for _ in range(a.warmup):
            step(x, y, *planes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step(x, y, *planes)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        return {"ms_per_step": round(ms, 2), "images_per_s": round(a.batch / ms * 1e3, 1),
                "peak_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "last_loss": float(loss)}

    out["native_kernels_graphed"] = graphed()
    hip_engine.native_training_convolutions = False  # the 3x3 convolutions back on torch's own kernels, everything else native
    try:
        out["native_kernels_graphed_torch_convolutions"] = graphed()
    finally:
        hip_engine.native_training_convolutions = True
    hip_engine.native_training_kernels = False   # the same step with torch's own attention / BatchNorm / InstanceNorm
    try:
        model = build(a.size, dev)
        ms, gib, loss = time_steps(model, data, du, a.steps, a.warmup)
        out["all_torch"] = {"ms_per_step": round(ms, 2), "images_per_s": round(a.batch / ms * 1e3, 1),
                            "peak_GiB": round(gib, 2), "last_loss": loss}
    except torch.OutOfMemoryError as e:
        out["all_torch"] = {"error": "out of memory: " + str(e)[:80]}
    finally:
        hip_engine.native_training_kernels = True
    L = (a.size // 2) ** 2
    out["core_alone"] = {f"B{a.batch}_heads4_L{L}": time_core(a.batch, 4, L)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
