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
PROF_FLASH_ATTENTION, PROF_IGEMM = 1, 2  # danra_profile_begin kinds (include/danra_hip.h)

WORKLOADS = {
    # the benchmark configurations: unconditional 64x64, conditional 64x64 (the headline), conditional 128x128
    "c2": dict(name="DANRA unconditional 64x64, T=1000, batch {batch}", size=64, batch=32, cond=False),
    "c3": dict(name="DANRA conditional (lsm+topo+cond image+season embedding) 64x64, T=1000, batch {batch}", size=64,
               batch=64, cond=True),
    "c4": dict(name="DANRA conditional 128x128, T=1000, batch {batch}", size=128, batch=32, cond=True),
    # the DDPM_clean_application variant (attention blocks with a feed-forward part); secondary workloads
    "clean": dict(name="clean-application network, conditional (lsm+topo+cond image+season) 64x64, T=1000, batch {batch}",
                  size=64, batch=64, cond=True, clean=True),
    "clean_h1": dict(name="clean-application network with its configuration default of 1 attention head, conditional "
                          "(lsm+topo+cond image+season) 64x64, T=1000, batch {batch}", size=64, batch=64, cond=True,
                     clean=True, heads=1),
    "clean_cfg": dict(name="clean-application network, season label only, classifier-free guidance 3.0 (two UNet "
                           "evaluations per step) 64x64, T=1000, batch {batch}", size=64, batch=64, cond=False, clean=True,
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


def library_sha256():
    """Hash of the library file this process runs (the one diffusionmodelscustom_amd.hip_engine loads)."""
    import hashlib
    from diffusionmodelscustom_amd import hip_engine
    with open(hip_engine.library_path(), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def family_traffic_per_launch(workload, batch, family):
    """HBM bytes per launch of the timed kernel family from the newest committed PMC measurement (rocprofv3 --pmc
    FETCH_SIZE with the gfx950 x2 correction, and WRITE_SIZE, in separate passes: tools/pmc_traffic.sh), averaged over
    the launches of a UNet evaluation like `achieved`.  It is a STATIC figure, measured once for the conditional 64x64
    workload (c3) at batch 64, so it is quoted only when the file names the library that is running now (sha256 of
    the .so written by tools/pmc_summary.py); otherwise `traffic` is null and `traffic_note` says why."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_family_traffic.json")))
    if workload != "c3" or batch != 64:
        return {"traffic": None, "traffic_note": "PMC traffic is only measured for workload c3 at batch 64"}
    if not files:
        return {"traffic": None, "traffic_note": "no committed PMC measurement"}
    with open(files[-1]) as fh:
        doc = json.load(fh)
    t = doc.get(family)
    name = os.path.relpath(files[-1], ROOT)
    if not t:
        return {"traffic": None, "traffic_note": f"{name} holds no entry for this kernel family"}
    if doc.get("library_sha256") != library_sha256():
        return {"traffic": None, "traffic_note": f"{name} was measured on another build of the library (sha256 "
                f"{str(doc.get('library_sha256'))[:16]}, running {library_sha256()[:16]}): regenerate it with tools/pmc_traffic.sh"}
    return {"traffic": t["bytes_per_launch"], "traffic_static": True,
            "traffic_unit": f"HBM bytes per launch: static PMC measurement of this library build ({name}; " + t["source"] + ")"}


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


def cpu_baseline(w, n_steps=16, n_steps_faithful=6, batch=None):
    """The CPU restatement (oracle/cpu_ref.py, parity-pinned to the reference) on the host cores, in both forms:
    `vectorised` (same arithmetic, the decoder's sinusoidal embedding as tensor expressions and attention without the
    averaged weights) and `faithful` (as the reference executes it: the embedding's Python loop over batch x 128
    scalar tensor operations, modules_DANRA_conditional.py:55-61, and need_weights attention, :105).  Each: denoise steps
    (UNet forward + update) at the workload's batch after one warm-up step, extrapolated to T-1 steps.  `value` is the
    faster form's."""
    from oracle import cpu_ref
    batch = batch or w["batch"]
    model = build_model(w, 42)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    inp = workload_inputs(w, batch, 42)
    clean, guide = bool(w.get("clean")), float(w.get("cfg_scale", 0.0))
    betas, alphas, alpha_hat = cpu_ref.schedule(1000, 1e-4, 0.02)
    heads = w.get("heads", 4)

    def run(steps, faithful):
        x = inp["x"]
        times = []
        with torch.no_grad():
            for k, i in enumerate(range(999, 999 - (steps + 1), -1)):
                t0 = time.perf_counter()
                t = torch.full((batch,), i, dtype=torch.long)
                eps = cpu_ref.unet_forward(sd, x, t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"],
                                           faithful=faithful, clean=clean, n_heads=heads)
                if guide > 0:
                    eps = torch.lerp(cpu_ref.unet_forward(sd, x, t, faithful=faithful, clean=clean, n_heads=heads), eps, guide)
                c1, c2, c3 = cpu_ref.update_coefficients(betas, alphas, alpha_hat, i)
                x = cpu_ref.ddpm_update(x, eps, torch.randn_like(x), c1, c2, c3)
                if k:
                    times.append(time.perf_counter() - t0)
        return sum(times) / len(times)

    per_step = {"vectorised": run(n_steps, False)}
    if not clean:  # the clean-application reference has no Python-loop embedding to be faithful to
        per_step["faithful"] = run(n_steps_faithful, True)
    forms = {k: {"value": batch / (v * 999), "s_per_step": v, "cores": torch.get_num_threads(),
                 "steps_timed": n_steps if k == "vectorised" else n_steps_faithful} for k, v in per_step.items()}
    best = max(forms, key=lambda k: forms[k]["value"])
    return {
        "value": forms[best]["value"],
        "unit": "images/sec",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "form": best,
        "forms": forms,
        "sample": f"denoise steps (UNet forward + update) at batch {batch} after 1 warm-up, extrapolated x999: "
                  + "; ".join(f"{k} {v['steps_timed']} steps at {v['s_per_step']:.3f} s/step" for k, v in forms.items())
                  + f"; host cpu_count={os.cpu_count()}",
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
    ap.add_argument("--no-strict-fp32", action="store_true", help="skip the fp32-instructions-only twin measurement")
    ap.add_argument("--strict-calls", type=int, default=3)
    ap.add_argument("--roofline-kernel", default="igemm", choices=["igemm", "flash"],
                    help="the kernel family timed for `roofline`: the implicit GEMMs (largest share of GPU time) or the "
                         "flash attention kernels")
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
        engine.profile_begin(PROF_IGEMM if args.roofline_kernel == "igemm" else PROF_FLASH_ATTENTION)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches, kern_flops = (0.0, 0, 0.0) if args.no_kernel_timing else engine.profile_end()
    kern_exec, kern_split = (0.0, 0.0) if args.no_kernel_timing else engine.profile_executed()
    kern_bytes = 0.0 if args.no_kernel_timing else engine.profile_bytes()
    assert torch.isfinite(out).all()

    # The same workload with every product on fp32 matrix instructions (danra_set_fp32_products(1): no fp16 split
    # anywhere), timed in the same process: one warm-up call (its graph is captured anew), then `strict_calls` calls.
    strict = None
    if world == 1 and not args.no_strict_fp32:
        engine.lib.danra_set_fp32_products(1)
        try:
            one_step()
            fence()
            ts = time.perf_counter()
            for _ in range(args.strict_calls):
                out32 = one_step()
            fence()
            strict_elapsed = time.perf_counter() - ts
            assert torch.isfinite(out32).all()
        finally:
            engine.lib.danra_set_fp32_products(0)
        strict = {"value": B * args.strict_calls / strict_elapsed, "unit": "images/sec",
                  "ms_per_step": 1e3 * strict_elapsed / args.strict_calls,
                  "unet_ms": 1e3 * strict_elapsed / args.strict_calls / (T - 1), "calls": args.strict_calls, "warmup": 1,
                  "dtype": "f32 (fp32 matrix instructions only: danra_set_fp32_products(1))"}

    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        images = world * B * args.steps
        in_ch = 4 if w["cond"] else 1
        fl_img = unet_flops_per_image(w["size"], in_ch, bool(w.get("clean"))) * (2 if w.get("cfg_scale", 0.0) > 0 else 1)
        value = images / elapsed
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
            "dtype": "f32 (fp16 split-operand products: 22-bit operands, fp32 accumulate)",
            "data": "synthetic (seeded inputs, deterministic random-init weights, in-kernel Philox noise)" +
                    (" REHEARSAL: ranks share GPUs, gloo backend, not a measurement" if rehearsal else ""),
            "config": {"workload": w["name"].format(batch=f"{B}/GPU x {world} GPU{'s' if world > 1 else ''} = global batch {world * B}"),
                       "timesteps": T, "batch_per_gpu": B, "global_batch": world * B,
                       "image": f"{w['size']}x{w['size']}", "parallelism": f"batch-sharded x{world}, weights broadcast "
                       f"once ({sent} bytes), no collective in the loop"},
            "unet_ms": 1e3 * elapsed / args.steps / (T - 1),
            "flops_per_image_per_denoise_step": fl_img,
            "whole_path_tflops": value * (T - 1) * fl_img / 1e12 / world,
            "whole_path_frac_of_fp32_peak": value * (T - 1) * fl_img / 1e12 / world / FP32_MATRIX_PEAK_TFLOPS,
        }
        if kern_ms > 0 and kern_launches:
            # The matrix instructions the family issued: a launch in the split-operand form executes three fp16 products
            # per algorithmic fp32 product (the head_dim-16 attention core 3.5), the rest are fp32 instructions.  The
            # roofline is the time the matrix cores would need for exactly those instructions at their dense peaks.
            seconds = kern_ms * 1e-3
            if args.roofline_kernel == "igemm":
                exec_fp16 = 3.0 * kern_split
                exec_fp32 = kern_exec - exec_fp16
                name = ("k_igemm family (the implicit GEMMs behind every convolution, transposed convolution and projection: "
                        "k_igemm with its split-K reduction kernels, k_gemm_k64, and the window-staged convolution kernels "
                        "k_conv_win and k_conv8s2)")
            else:
                hd16 = w.get("heads", 4) == 4
                exec_fp16 = kern_exec if hd16 else 0.0
                exec_fp32 = 0.0 if hd16 else kern_exec
                name = "k_flash_hd16 (head_dim-16 flash attention)" if hd16 else "k_flash_wide (flash attention, head_dim 64/32)"
            peak_seconds = exec_fp16 / (FP16_MATRIX_PEAK_TFLOPS * 1e12) + exec_fp32 / (FP32_MATRIX_PEAK_TFLOPS * 1e12)
            executed_tflops = kern_exec / seconds / 1e12
            blended_peak = kern_exec / peak_seconds / 1e12
            frac = peak_seconds / seconds
            assert frac <= 1.0, f"roofline fraction {frac} above 1: the peak does not bound the kernel it is quoted for"
            line["roofline"] = {
                "kernel": name + ": HIP events attached to every dispatch of the family (they carry the kernel's own begin and end times) in every 64th denoise step of the timed region",
                "bound": "mfma",
                "achieved": executed_tflops,
                "peak": blended_peak,
                "unit": "TFLOP/s",
                "frac": frac,
                "traffic": None,
                "what": "achieved = FLOPs executed by the issued matrix instructions / kernel time; peak = the same FLOPs / the "
                        "time the matrix cores need for them at their dense peaks (fp16 2500, fp32 157.3 TFLOP/s)",
                "launches": kern_launches,
                "avg_launch_us": 1e3 * kern_ms / kern_launches,
                "algorithmic_flops_per_launch": kern_flops / kern_launches,
                "executed_flops_per_launch": kern_exec / kern_launches,
                "executed_fp16_share": exec_fp16 / kern_exec if kern_exec else None,
                "algorithmic_tflops": kern_flops / seconds / 1e12,
                "algorithmic_frac_of_fp32_peak": kern_flops / seconds / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            }
            line["roofline"]["algorithmic_bytes_per_launch"] = kern_bytes / kern_launches
            line["roofline"].update(family_traffic_per_launch(args.workload, B, args.roofline_kernel))
        if strict:
            line["strict_fp32"] = strict
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
