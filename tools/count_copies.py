"""Diagnostic: which tensor copies does one eager training step make (operator, shapes, how often)?"""
import os, sys
from collections import Counter
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.train_step_bench import build
from diffusionmodelscustom_amd.synthetic import synthetic_inputs

dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
B, size = 32, 64
d = synthetic_inputs(B, size, size, seed=42, conditional=True)
x, y = d["x"].to(dev), d["y"].to(dev)
planes = [d["cond_img"].to(dev), d["lsm_cond"].to(dev), d["topo_cond"].to(dev)]
m = build(size, dev)
t = torch.randint(1, 1000, (B,), device=dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, capturable=dev.type == "cuda")
def step():
    for p in m.parameters(): p.grad = None
    torch.nn.functional.mse_loss(m(x, t, y, *planes), x).backward()
    opt.step()
step(); step()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    step()
c = Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous"):
        c[(e.name, str(e.input_shapes)[:90])] += 1
tot = Counter()
for (n, s), k in c.items(): tot[n] += k
print(dict(tot))
for (n, s), k in sorted(c.items(), key=lambda kv: -kv[1])[:12]:
    print(k, n, s)
