"""Runs on the GPU box: repeated full-size sampling calls with changing batch / image size / T, watching device memory.
Prints the free-memory trace; fails if memory keeps shrinking after the shapes have all been seen once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, build_model, workload_inputs
from diffusionmodelscustom_amd import DiffusionUtils

dev = torch.device("cuda:0")
model = build_model(WORKLOADS["c3"], 42).to(dev)
shapes = [(64, 64, 300), (16, 64, 200), (8, 128, 100), (64, 64, 300), (3, 96, 50), (16, 64, 200)]
trace = []
for rep in range(3):
    for B, size, T in shapes:
        w = dict(WORKLOADS["c3"], size=size)
        inp = {k: (None if v is None else v.to(dev)) for k, v in workload_inputs(w, B, 1).items()}
        du = DiffusionUtils(T, 1e-4, 0.02, dev, "linear")
        du.noise_seed = 5
        out = du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        free, total = torch.cuda.mem_get_info()
        trace.append(free >> 20)
    print("pass", rep, "free MiB after each shape:", trace[-len(shapes):], flush=True)
n = len(shapes)
assert min(trace[2 * n:]) >= min(trace[n:2 * n]) - 64, "device memory keeps shrinking"
print("soak ok")
