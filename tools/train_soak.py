"""A few hundred replays of the graphed training step on one synthetic batch, with the library's split-operand fp16 products
and with fp32 matrix instructions only: both losses must stay finite, fall, and stay close to one another.

    python tools/train_soak.py [--steps 300]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(fp32_products, steps):
    from diffusionmodelscustom_amd import DiffusionUtils, hip_engine
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    from diffusionmodelscustom_amd.training import GraphedTrainStep
    from tools.train_step_bench import build
    lib = hip_engine.load_library()
    lib.danra_set_fp32_products(fp32_products)
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        d = synthetic_inputs(32, 64, 64, seed=42, conditional=True)
        b = dict(images=d["x"].to(dev), seasons=d["y"].to(dev), cond_images=d["cond_img"].to(dev), lsm=d["lsm_cond"].to(dev),
                 topo=d["topo_cond"].to(dev))
        model = build(64, dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, fused=True)
        step = GraphedTrainStep(model, opt, DiffusionUtils(1000, 1e-4, 0.02, device=dev), torch.nn.MSELoss(), b)
        losses = []
        for i in range(steps):
            loss = step(**b)
            if i % 10 == 9 or i == 0:
                losses.append(float(loss.item()))
        return losses
    finally:
        lib.danra_set_fp32_products(0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    split, strict = run(0, a.steps), run(1, a.steps)
    ok = all(x == x and abs(x) < 1e6 for x in split + strict)
    print(json.dumps({"steps": a.steps, "split_operand": [round(x, 4) for x in split], "fp32_instructions": [round(x, 4) for x in strict],
                      "finite": ok, "fell": split[-1] < 0.8 * split[0] and strict[-1] < 0.8 * strict[0],
                      "largest_relative_gap": round(max(abs(x - y) / max(abs(y), 1e-9) for x, y in zip(split, strict)), 4)}))
