"""Which torch operators issue the copy kernels of one eager training step: torch.profiler over a few steps, grouped by
operator name and input shapes (top entries by self device time).

    python tools/prof_train_ops.py [--size 64] [--batch 32]
"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusionmodelscustom_amd import DiffusionUtils  # noqa: E402
from diffusionmodelscustom_amd.synthetic import synthetic_inputs  # noqa: E402
from tools.train_step_bench import build  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--copies", action="store_true", help="attribute the aten::copy_ calls to python frames instead")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    d = synthetic_inputs(args.batch, args.size, args.size, seed=42, conditional=True)
    x, y = d["x"].to(dev), d["y"].to(dev)
    planes = [d["cond_img"].to(dev), d["lsm_cond"].to(dev), d["topo_cond"].to(dev)]
    model = build(args.size, dev)
    du = DiffusionUtils(1000, 1e-4, 0.02, device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    loss_fn = torch.nn.MSELoss()

    def step():
        t = du.sampleTimesteps(x.shape[0])
        noisy, noise = du.noiseImage(x, t)
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(noisy, t, y, *planes), noise)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=args.copies) as prof:
        step()
        torch.cuda.synchronize()
    if not args.copies:
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40,
                                                                 max_name_column_width=60, max_shapes_column_width=70))
        return
    # who asks for the copies: aten::copy_ calls grouped by the innermost frames of this package / torch.nn on their stack
    from collections import defaultdict
    groups = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name != "aten::copy_" or ev.device_time_total <= 0:
            continue
        frames = [f for f in (ev.stack or []) if "diffusionmodelscustom_amd" in f or "torch/nn/modules" in f or "autograd" in f]
        key = " <- ".join(f.split("/")[-1][:70] for f in frames[:3]) or "(no python frame: backward engine)"
        key += "   " + str(ev.input_shapes[:2])
        groups[key][0] += 1
        groups[key][1] += ev.device_time_total
    for key, (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{us:9.1f} us  x{n:<4d} {key}")


if __name__ == "__main__":
    main()
