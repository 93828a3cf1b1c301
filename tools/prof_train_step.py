"""Per-kernel time of ONE graphed training step: run under `rocprofv3 --kernel-trace --output-format csv`, then pass the
trace to this script with --trace to list the kernels of the last replay.

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 tools/prof_train_step.py --size 64 --batch 32
    python3 tools/prof_train_step.py --trace out/t_kernel_trace.csv
"""
import argparse
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(size, batch, replays):
    import torch
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.synthetic import synthetic_inputs
    from diffusionmodelscustom_amd.training import GraphedTrainStep
    from tools.train_step_bench import build
    dev = torch.device("cuda:0")
    d = synthetic_inputs(batch, size, size, seed=42, conditional=True)
    b = dict(images=d["x"].to(dev), seasons=d["y"].to(dev), cond_images=d["cond_img"].to(dev), lsm=d["lsm_cond"].to(dev),
             topo=d["topo_cond"].to(dev))
    model = build(size, dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, fused=True)  # as tools/train_step_bench.py
    step = GraphedTrainStep(model, opt, DiffusionUtils(1000, 1e-4, 0.02, device=dev), torch.nn.MSELoss(), b)
    for _ in range(replays):
        step(**b)
        torch.cuda.synchronize()


def summarise(path, top, sequence=False, width=100):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the replays run the same kernel sequence: the last replay = the shortest tail that repeats the tail before it
    names = [r["Kernel_Name"] for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    starts = [int(r["Start_Timestamp"]) for r in rows]
    cut = 0
    for n in range(50, len(names) // 2):
        if names[-n:] == names[-2 * n:-n]:
            cut = len(names) - n
            break
    last = rows[cut:]
    total = defaultdict(lambda: [0, 0])
    for r in last:
        t = total[r["Kernel_Name"][:width]]
        t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        t[1] += 1
    busy = sum(v[0] for v in total.values())
    print(f"last replay: {len(last)} kernels, busy {busy / 1e3:.0f} us, wall {(ends[-1] - starts[cut]) / 1e3:.0f} us")
    for name, (ns, n) in sorted(total.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{ns / 1e3:9.1f} us  x{n:<4d} {100 * ns / busy:5.1f} %  {name}")
    if sequence:   # the replay's kernels in launch order, runs of one name collapsed (time of the run, grid of its last launch)
        print("--- launch order")
        run_name, run_n, run_ns, grid = None, 0, 0, ""
        for r in last + [None]:
            name = None if r is None else r["Kernel_Name"][:70]
            if name != run_name:
                if run_name is not None:
                    print(f"  x{run_n:<3d} {run_ns / 1e3:7.1f} us  {grid:>12s}  {run_name}")
                run_name, run_n, run_ns = name, 0, 0
            if r is not None:
                run_n += 1
                run_ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                wg = max(int(r.get("Workgroup_Size_X", 1) or 1), 1)
                grid = "x".join(str(int(r.get(f"Grid_Size_{a}", 1) or 1) // (wg if a == "X" else 1)) for a in "XYZ")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--replays", type=int, default=4)
    ap.add_argument("--trace")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--sequence", action="store_true")
    ap.add_argument("--width", type=int, default=100, help="characters of a kernel name that tell kernels apart")
    a = ap.parse_args()
    if a.trace:
        summarise(a.trace, a.top, a.sequence, a.width)
    else:
        run(a.size, a.batch, a.replays)
