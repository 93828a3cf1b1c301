"""Host-side cost of the eager training step (the reference's loop body as written, one Python call per operator): times ten
steps with the native convolutions on and off, then lists where the host time of five steps goes (cProfile).
Runs on the GPU box:  python tools/prof_eager_step.py"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionmodelscustom_amd import DiffusionUtils, hip_engine
from diffusionmodelscustom_amd.synthetic import synthetic_inputs
from tools.train_step_bench import build
dev = torch.device("cuda:0")
d = synthetic_inputs(32, 64, 64, seed=42, conditional=True)
x, y, planes = d["x"].to(dev), d["y"].to(dev), [d["cond_img"].to(dev), d["lsm_cond"].to(dev), d["topo_cond"].to(dev)]
du = DiffusionUtils(1000, 1e-4, 0.02, device=dev)
model = build(64, dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
def one():
    t = du.sampleTimesteps(x.shape[0])
    noisy, noise = du.noiseImage(x, t)
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(model(noisy, t, y, *planes), noise)
    loss.backward()
    opt.step()
for _ in range(3): one()
torch.cuda.synchronize()
for flag in (True, False):
    hip_engine.native_training_convolutions = flag
    for _ in range(2): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): one()
    torch.cuda.synchronize()
    print("native convs", flag, (time.perf_counter() - t0) / 10 * 1e3, "ms")
hip_engine.native_training_convolutions = True
pr = cProfile.Profile()
pr.enable()
for _ in range(5): one()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
