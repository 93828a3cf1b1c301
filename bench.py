"""Benchmark: sampled images/sec of the 1000-step DDPM sampler (DiffusionUtils.sample through the HIP
path) on N MI355X GPUs of one node, with the CPU restatement timed beside it.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one whole sample() call: T-1 = 999 UNet evaluations + 999 DDPM updates for one batch of
synthetic conditional inputs per GPU (weak scaling: every GPU samples its own batch of the same size;
no collective inside the loop).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffusionmodelscustom_amd import Decoder, DiffusionNet, DiffusionUtils, Encoder  # noqa: E402
from diffusionmodelscustom_amd.distributed import broadcast_model_  # noqa: E402
from diffusionmodelscustom_amd.synthetic import fill_state_dict_, synthetic_inputs  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix/vector peak
FP16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X dense fp16 matrix peak
# k_flash_hd16 evaluates its two fp32 products as fp16 products of operands split in two halves: per 32 keys and
# 16-query tile 4 + 3 instructions of 16x16x32 multiply-adds against 2 x 16x32x16 algorithmic ones
FLASH_HD16_EXECUTED_OVER_ALGORITHMIC = (4 + 3) * 16 * 16 * 32 / (2.0 * 16 * 32 * 16)

WORKLOADS = {
    # the benchmark configurations: unconditional 64x64, conditional 64x64 (the headline), conditional 128x128
    "c2": dict(name="DANRA unconditional 64x64, T=1000, batch 32/GPU", size=64, batch=32, cond=False),
    "c3": dict(name="DANRA conditional (lsm+topo+cond image+season embedding) 64x64, T=1000, batch 64/GPU", size=64,
               batch=64, cond=True),
    "c4": dict(name="DANRA conditional 128x128, T=1000, batch 32/GPU", size=128, batch=32, cond=True),
    # the DDPM_clean_application variant (attention blocks with a feed-forward part); secondary workloads
    "clean": dict(name="clean-application network, conditional (lsm+topo+cond image+season) 64x64, T=1000, batch 64/GPU",
                  size=64, batch=64, cond=True, clean=True),
    "clean_h1": dict(name="clean-application network with its configuration default of 1 attention head, conditional "
                          "(lsm+topo+cond image+season) 64x64, T=1000, batch 64/GPU", size=64, batch=64, cond=True,
                     clean=True, heads=1),
    "clean_cfg": dict(name="clean-application network, season label only, classifier-free guidance 3.0 (two UNet "
                           "evaluations per step) 64x64, T=1000, batch 64/GPU", size=64, batch=64, cond=False, clean=True,
                      labels=True, cfg_scale=3.0),
}


def unet_flops_per_image(size, in_ch, attn_ff=False):
    """2 x multiply-adds of every convolution, linear and attention product of one UNet evaluation
    (norms, activations and the softmax itself excluded)."""
    fl = 0.0
    s2, s4 = size // 2, size // 4

    def attn(L, C):
        return 2.0 * L * C * 3 * C + 4.0 * L * L * C + 2.0 * L * C * C + (4.0 * L * C * C if attn_ff else 0.0)

    fl += 2.0 * s2 * s2 * 64 * 64 * in_ch            # conv1 8x8
    fl += attn(s2 * s2, 64)
    fl += 2.0 * s4 * s4 * 64 * 64 * 64               # conv2 8x8
    side, cin = s4, 64
    for lvl, c in enumerate((64, 128, 256, 512)):
        if lvl:
            side //= 2
        hw = side * side
        fl += 2.0 * hw * c * cin * 9 + 3 * 2.0 * hw * c * c * 9   # two BasicBlocks
        if lvl:
            fl += 2.0 * hw * c * cin                               # 1x1 shortcut
        fl += attn(hw, c)
        cin = c
    for cin, cout in ((512, 256), (256, 128), (128, 64), (64, 64)):
        side *= 2
        hw = side * side
        fl += 2.0 * hw * cin * cin + 2.0 * hw * cout * cin * 9    # convT (one tap per output pixel) + conv3x3
        fl += attn(hw, cout)
    side *= 2
    fl += 2.0 * side * side * 64 * 64 + 2.0 * side * side * 64 * 9
    fl += 2.0 * 256 * 1536                                        # time projections
    return fl


def flash_traffic_per_launch(workload, batch):
    """HBM bytes per k_flash_hd16 launch from the committed PMC measurement (rocprofv3 --pmc FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE, separate passes: tools/pmc_traffic.sh), averaged like `achieved` over the
    launches of one UNet evaluation (two at L=1024, two at L=256).  Only measured for the conditional 64x64 workload (c3) at batch 64."""
    path = os.path.join(ROOT, "profiles", "r01_flash_traffic.json")
    if workload != "c3" or batch != 64 or not os.path.exists(path):
        return None
    with open(path) as fh:
        t = json.load(fh)["k_flash_hd16"]
    per = [t[k]["read_bytes"] + t[k]["write_bytes"] for k in ("L1024_B64", "L256_B64")]
    return sum(per) / len(per)


def build_model(w, seed):
    size = w["size"]
    if w.get("clean"):
        from diffusionmodelscustom_amd import unet
        enc = unet.Encoder(1, 256, cond_on_lsm=w["cond"], cond_on_topo=w["cond"], cond_on_img=w["cond"],
                           cond_img_dim=(1, size, size) if w["cond"] else None, block_layers=[2, 2, 2, 2],
                           num_classes=4 if (w["cond"] or w.get("labels")) else None, n_heads=w.get("heads", 4))
        dec = unet.Decoder(512, 1, 256, 64, n_heads=w.get("heads", 4))
        return fill_state_dict_(unet.DiffusionNet(enc, dec), seed).eval()
    if w["cond"]:
        enc = Encoder(1, 256, lsm_tensor=torch.zeros(1, size, size), topo_tensor=torch.zeros(1, size, size),
                      cond_on_img=True, cond_img_dim=(1, size, size), block_layers=[2, 2, 2, 2], num_classes=4)
    else:
        enc = Encoder(1, 256, block_layers=[2, 2, 2, 2])
    model = DiffusionNet(enc, Decoder(512, 1, 256, 64))
    return fill_state_dict_(model, seed).eval()


def workload_inputs(w, batch, seed):
    inp = synthetic_inputs(batch, w["size"], w["size"], seed=seed, conditional=w["cond"] or bool(w.get("labels")))
    if not w["cond"]:
        inp.update(cond_img=None, lsm_cond=None, topo_cond=None)
    return inp


def cpu_baseline(w, n_steps=25, batch=None):
    """The CPU restatement (oracle/cpu_ref.py, parity-pinned to the reference) on the host cores: `n_steps`
    denoise steps (UNet + update) at the workload's batch after one warm-up step (about 10-20 s of CPU work),
    extrapolated to T-1 steps."""
    from oracle import cpu_ref
    batch = batch or w["batch"]
    model = build_model(w, 42)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    inp = workload_inputs(w, batch, 42)
    clean, guide = bool(w.get("clean")), float(w.get("cfg_scale", 0.0))
    betas, alphas, alpha_hat = cpu_ref.schedule(1000, 1e-4, 0.02)
    x = inp["x"]
    times = []
    with torch.no_grad():
        for k, i in enumerate(range(999, 999 - (n_steps + 1), -1)):
            t0 = time.perf_counter()
            t = torch.full((batch,), i, dtype=torch.long)
            heads = w.get("heads", 4)
            eps = cpu_ref.unet_forward(sd, x, t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], clean=clean,
                                       n_heads=heads)
            if guide > 0:
                eps = torch.lerp(cpu_ref.unet_forward(sd, x, t, clean=clean, n_heads=heads), eps, guide)
            c1, c2, c3 = cpu_ref.update_coefficients(betas, alphas, alpha_hat, i)
            x = cpu_ref.ddpm_update(x, eps, torch.randn_like(x), c1, c2, c3)
            if k:
                times.append(time.perf_counter() - t0)
    per_step = sum(times) / len(times)
    return {
        "value": batch / (per_step * 999),
        "unit": "images/sec",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{n_steps} denoise steps (UNet forward + update) at batch {batch} after 1 warm-up, "
                  f"{per_step:.3f} s/step, extrapolated x999; vectorised time embedding (the faster of the two "
                  f"CPU forms); host cpu_count={os.cpu_count()}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    # Rehearsal of the multi-process plumbing on a box with fewer GPUs than ranks (DANRA_BENCH_REHEARSAL=1): ranks
    # share the visible GPUs and talk over gloo.  Not a measurement; the line it prints says so.
    rehearsal = os.environ.get("DANRA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    w = dict(WORKLOADS[args.workload])
    B = args.batch or w["batch"]
    T = args.timesteps
    # rank 0 holds the weights; the other ranks start from a different fill and receive them by broadcast
    model = build_model(w, 42 if rank == 0 else 1000 + rank).to(dev)
    sent = broadcast_model_(model, src=0)
    inp = workload_inputs(w, B, 42 + rank)
    inp = {k: (None if v is None else v.to(dev)) for k, v in inp.items()}
    if w.get("clean"):
        from diffusionmodelscustom_amd.diffusion_modules import DiffusionUtils as CleanDiffusionUtils
        du = CleanDiffusionUtils(T, 1e-4, 0.02, dev, "linear", img_size=w["size"])
    else:
        du = DiffusionUtils(T, 1e-4, 0.02, dev, "linear")
    du.noise_seed = 42 + rank
    engine = model.hip_engine(dev)

    def one_step():
        if w.get("clean"):  # draws its own start image on the host, as the reference does
            out = du.sample(B, model, 1, y=inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                            topo_cond=inp["topo_cond"], cfg_scale=w.get("cfg_scale", 0.0))
            model.eval()
            return out
        return du.sample(inp["x"], model, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                         topo_cond=inp["topo_cond"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    fence()
    if not args.no_kernel_timing:
        engine.profile_begin(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches, kern_flops = (0.0, 0, 0.0) if args.no_kernel_timing else engine.profile_end()
    assert torch.isfinite(out).all()

    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        images = world * B * args.steps
        in_ch = 4 if w["cond"] else 1
        fl_img = unet_flops_per_image(w["size"], in_ch, bool(w.get("clean"))) * (2 if w.get("cfg_scale", 0.0) > 0 else 1)
        value = images / elapsed
        achieved = (kern_flops / (kern_ms * 1e-3) / 1e12) if kern_ms > 0 else None
        line = {
            "metric": "sampled images/sec (1000-step DDPM, 64x64) at 1/2/4/8 MI355X vs CPU ref",
            "value": value,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded inputs, deterministic random-init weights, in-kernel Philox noise)" +
                    (" REHEARSAL: ranks share GPUs, gloo backend, not a measurement" if rehearsal else ""),
            "config": {"workload": w["name"], "timesteps": T, "batch_per_gpu": B, "global_batch": world * B,
                       "image": f"{w['size']}x{w['size']}", "parallelism": f"batch-sharded x{world}, weights broadcast "
                       f"once ({sent} bytes), no collective in the loop"},
            "unet_ms": 1e3 * elapsed / args.steps / (T - 1),
            "flops_per_image_per_denoise_step": fl_img,
            "whole_path_tflops": value * (T - 1) * fl_img / 1e12 / world,
            "whole_path_frac_of_fp32_peak": value * (T - 1) * fl_img / 1e12 / world / FP32_MATRIX_PEAK_TFLOPS,
            "roofline": {
                "kernel": ("k_flash_hd16 (head_dim-16 flash attention" if w.get("heads", 4) == 4 else
                           "k_flash_wide (flash attention, head_dim 64/32") +
                          ": HIP events around its launches in every 8th denoise step of the timed region)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP32_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (achieved / FP32_MATRIX_PEAK_TFLOPS) if achieved else None,
                "traffic": flash_traffic_per_launch(args.workload, B),
                "traffic_unit": "bytes per launch (PMC, profiles/r01_pmc_traffic.txt; algorithmic Q+K+V+O = "
                                f"{(4 * 64 * 64 * (1024 + 256) * 4) / 2:.0f})" if (args.workload == "c3" and B == 64) else None,
                "launches": kern_launches,
                "avg_launch_us": (1e3 * kern_ms / kern_launches) if kern_launches else None,
                "algorithmic_flops_per_launch": (kern_flops / kern_launches) if kern_launches else None,
            },
        }
        if achieved and w.get("heads", 4) == 4:
            # `achieved` / `peak` above: algorithmic fp32 FLOP/s against the fp32 matrix peak (what an all-fp32 kernel is
            # bounded by).  The kernel's instructions are fp16 matrix instructions on split operands; against THEIR peak:
            executed = achieved * FLASH_HD16_EXECUTED_OVER_ALGORITHMIC
            line["roofline"]["instructions"] = ("v_mfma_f32_16x16x32_f16 on operands split in two fp16 halves (22 bits "
                                                "each, fp32 accumulation); %.1fx the algorithmic multiply-adds are "
                                                "executed" % FLASH_HD16_EXECUTED_OVER_ALGORITHMIC)
            line["roofline"]["executed"] = executed
            line["roofline"]["executed_peak"] = FP16_MATRIX_PEAK_TFLOPS
            line["roofline"]["executed_frac"] = executed / FP16_MATRIX_PEAK_TFLOPS
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
