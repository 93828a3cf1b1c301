"""One training step's parameter gradients, evaluated five ways and compared with a float64 evaluation on the CPU:

  native-split : GPU, the 64-channel attention cores in libdanra_hip.so (split-operand fp16 products, the default)
  native-fp32  : GPU, the same kernels multiplying with fp32 matrix instructions only
  torch-gpu    : GPU, torch's own attention
  torch-cpu    : CPU, float32
  float64      : CPU, float64 (the arbiter)

Prints, per evaluation, the rms error of every gradient tensor against float64 relative to that tensor's rms (floored at
1 % of the largest tensor rms, as the test does), the worst tensors, and the stem's weight gradient in particular.

    python tools/diag_train_grad.py [--batch 2] [--model clean|danra]
"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diffusionmodelscustom_amd import hip_engine  # noqa: E402
from diffusionmodelscustom_amd import modules_DANRA_conditional as mods  # noqa: E402
from tests.model_util import build_clean_model, build_model  # noqa: E402


def as_double(model):
    """A float64 copy; the two sinusoidal embeddings stay float32 expressions (inputs of the network, evaluated as every
    other path evaluates them) and are cast on the way out."""
    m = copy.deepcopy(model).double()
    for mod in m.modules():
        if isinstance(mod, mods.SinusoidalEmbedding):
            fwd = mod.forward
            mod.forward = (lambda f: (lambda x: f(x).double()))(fwd)
    enc = m.encoder
    pe = enc.pos_encoding
    enc.pos_encoding = lambda t, ch: pe(t, ch).double()
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--model", default="clean")
    ap.add_argument("--seed", type=int, default=11)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    model = (build_clean_model if args.model == "clean" else build_model)()
    model.train()
    B = args.batch
    x = torch.randn(B, 1, 64, 64)
    t = torch.tensor([17, 640] * (B // 2) + [333] * (B % 2))
    y = torch.tensor([1, 3] * (B // 2) + [0] * (B % 2))
    planes = [torch.randn(B, 1, 64, 64) for _ in range(3)]
    target = torch.randn(B, 1, 64, 64)

    def step(m, dev, dtype=torch.float32):
        m.zero_grad()
        pred = m(x.to(dev, dtype), t.to(dev), y.to(dev), *(p.to(dev, dtype) for p in planes))
        loss = torch.nn.functional.mse_loss(pred, target.to(dev, dtype))
        loss.backward()
        return loss.item(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}

    lib = hip_engine.load_library()
    results = {}
    results["float64"] = step(as_double(model), "cpu", torch.float64)
    results["torch-cpu"] = step(copy.deepcopy(model), "cpu")
    gpu = copy.deepcopy(model).to("cuda:0")
    lib.danra_set_fp32_products(0)
    results["native-split"] = step(gpu, "cuda:0")
    lib.danra_set_fp32_products(1)
    results["native-fp32"] = step(gpu, "cuda:0")
    lib.danra_set_fp32_products(0)
    supported = hip_engine.attention_core_supported
    hip_engine.attention_core_supported = lambda *a: False
    results["torch-gpu"] = step(gpu, "cuda:0")
    # everything torch: BatchNorm / InstanceNorm kernels off as well
    hip_engine.native_training_kernels = False
    results["alltorch-gpu"] = step(gpu, "cuda:0")
    hip_engine.native_training_kernels = True
    hip_engine.attention_core_supported = supported

    ref_loss, ref = results["float64"]
    rms = {n: g.norm().item() / g.numel() ** 0.5 for n, g in ref.items()}
    floor = 1e-2 * max(rms.values())
    print(f"model={args.model} batch={B}  float64 loss {ref_loss:.9f}  largest tensor rms {max(rms.values()):.3e}")
    for name, (loss, grads) in results.items():
        if name == "float64":
            continue
        rel = {n: ((grads[n] - ref[n]).norm().item() / ref[n].numel() ** 0.5) / max(rms[n], floor) for n in ref}
        worst = sorted(rel, key=rel.get, reverse=True)[:4]
        print(f"{name:13s} loss rel err {abs(loss - ref_loss) / abs(ref_loss):.2e}   encoder.conv1.weight "
              f"{rel['encoder.conv1.weight']:.3e}   worst: " + ", ".join(f"{n} {rel[n]:.2e}" for n in worst))
    # pairwise on the stem gradient, as the test compares them
    names = [n for n in results if n != "float64"]
    key = "encoder.conv1.weight"
    print(f"pairwise rms differences of {key} relative to its rms {rms[key]:.3e}:")
    for i, a in enumerate(names):
        row = []
        for b in names:
            d = (results[a][1][key] - results[b][1][key]).norm().item() / ref[key].numel() ** 0.5 / max(rms[key], floor)
            row.append(f"{d:.2e}")
        print(f"  {a:13s} " + "  ".join(row))


if __name__ == "__main__":
    main()
