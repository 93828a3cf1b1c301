"""Experiment (GPU box): one sampling call at batch 64 against two concurrent calls at batch 32 on two streams
(two model instances = two native handles and workspaces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, build_model, workload_inputs
from diffusionmodelscustom_amd import DiffusionUtils

dev = torch.device("cuda:0")
w = WORKLOADS["c3"]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
models = [build_model(w, 42).to(dev) for _ in range(parts)]
inp = {k: v.to(dev) for k, v in workload_inputs(w, 64, 42).items()}
du = DiffusionUtils(T, 1e-4, 0.02, dev, "linear")
du.noise_seed = 1
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

def whole():
    return du.sample(inp["x"], models[0], inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])

def split():
    outs = []
    n = 64 // parts
    for i, (m, s) in enumerate(zip(models, streams)):
        sl = slice(i * n, (i + 1) * n)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append(du.sample(inp["x"][sl], m, inp["y"][sl], cond_img=inp["cond_img"][sl], lsm_cond=inp["lsm_cond"][sl],
                                  topo_cond=inp["topo_cond"][sl]))
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    return torch.cat(outs)

for name, fn in (("whole", whole), ("split", split), ("whole", whole), ("split", split)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {1e3 * dt / (T - 1):.4f} ms per denoise step of 64 images, finite={bool(torch.isfinite(out).all())}", flush=True)
