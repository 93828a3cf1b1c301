"""The differentiable attention core (danra_attention_core_fwd / _bwd) against torch's own arithmetic.

The checker is the reference's composition - nn.MultiheadAttention as ImageSelfAttention calls it
(modules_DANRA_conditional.py:91-110) and its autograd backward - evaluated on the CPU (float64 for the bare
core, the float32 module for the block and the network).  Tolerances: 1e-5 * max|.| per tensor for the core and the
block (the per-op bar of the inference kernels); the whole-network test states its own.
"""
import copy
import ctypes

import pytest
import torch

from diffusionmodelscustom_amd import hip_engine
from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream
from diffusionmodelscustom_amd.modules_DANRA_conditional import ImageSelfAttention
from tests.model_util import build_clean_model, build_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def core_reference(q, k, v, dout):
    """float64 softmax(q k^T / 4) v and its gradients; q, k, v: [B, heads, L, 16]; dout tokens [B, L, heads*16]."""
    q, k, v = (t.double().requires_grad_(True) for t in (q, k, v))
    p = torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1)
    out = (p @ v).permute(0, 2, 1, 3).reshape(q.shape[0], q.shape[2], -1)
    out.backward(dout.double())
    return out.detach(), q.grad, k.grad, v.grad


def run_core(q, k, v, dout):
    lib = hip_engine.load_library()
    B, heads, L, hd = q.shape
    qd, kd, vd, dd = (t.to(DEV).contiguous() for t in (q, k, v, dout))
    out = torch.empty((B, L, heads * hd), device=DEV)
    lse2 = torch.empty((B * heads, L), device=DEV)
    delta = torch.empty_like(lse2)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    _check(lib, lib.danra_attention_core_fwd(_ptr(qd), _ptr(kd), _ptr(vd), _ptr(out), _ptr(lse2), B, heads, L, hd,
                                             _stream()), "fwd")
    _check(lib, lib.danra_attention_core_bwd(_ptr(qd), _ptr(kd), _ptr(vd), _ptr(out), _ptr(dd), _ptr(lse2), _ptr(dq),
                                             _ptr(dk), _ptr(dv), _ptr(delta), B, heads, L, hd, _stream()), "bwd")
    torch.cuda.synchronize()
    return out.cpu(), dq.cpu(), dk.cpu(), dv.cpu(), lse2.cpu()


def close(got, want, tol, what):
    err = (got.double() - want.double()).abs().max().item()
    ref = want.double().abs().max().item()
    assert err <= tol * ref, f"{what}: max error {err:.3e} against max |.| {ref:.3e} (tolerance {tol:g} relative)"


@pytest.mark.parametrize("B,heads,L,qscale", [
    (2, 4, 256, 1.0),    # B*heads = 8: the XCD-grouped block order
    (3, 4, 1024, 1.0),   # B*heads = 12: the plain block order
    (1, 2, 4096, 1.0),   # the 128x128 network's first level
    (2, 4, 256, 12.0),   # peaked rows: scores up to a few hundred, one key takes nearly all the weight
])
def test_core_forward_and_backward_match_float64(B, heads, L, qscale):
    g = torch.Generator().manual_seed(1234 + L)
    q = torch.randn(B, heads, L, 16, generator=g) * qscale
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g) * 2 + 0.5
    dout = torch.randn(B, L, heads * 16, generator=g)
    want_out, want_dq, want_dk, want_dv = core_reference(q, k, v, dout)
    out, dq, dk, dv, lse2 = run_core(q, k, v, dout)
    close(out, want_out, 1e-5, "out")
    close(dv, want_dv, 1e-5, "dv")
    close(dq, want_dq, 1e-5, "dq")
    close(dk, want_dk, 1e-5, "dk")
    want_lse2 = torch.logsumexp(q.double() @ k.double().transpose(-1, -2) / 4.0, dim=-1) * 1.4426950408889634
    close(lse2.reshape(B, heads, L), want_lse2, 1e-6, "lse2")


@pytest.mark.parametrize("case", ["huge_values_late", "huge_values_first_tile", "tiny_values", "wide_probability_range",
                                  "nonfinite_value", "huge_keys_late", "huge_queries", "tiny_keys_large_queries"])
def test_core_forward_value_ranges(case):
    """The forward kernel multiplies K Q^T and P V on the fp16 matrix cores (operands split in fp16 halves) while every
    |K|, |V| it has staged and every scaled |Q| of the workgroup is below 2^15, and in fp32 from the first tile on that
    holds a larger (or non-finite) value.  Both forms, and the switch between them in the middle of a row, stay within
    the core's tolerance of float64."""
    B, heads, L = 2, 4, 1024
    g = torch.Generator().manual_seed(99)
    q = torch.randn(B, heads, L, 16, generator=g)
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g)
    if case == "huge_values_late":
        v[:, 1::2, 640:] *= 3e5          # every second head switches to fp32 products at key 640
    elif case == "huge_values_first_tile":
        v[:, :, 3, 5] = -7e6
    elif case == "tiny_values":
        v *= 1e-6                        # fp16 subnormals in both halves
    elif case == "wide_probability_range":
        q *= 6.0                         # probabilities from 1 down to far below the fp16 range
        v[:, :, ::7] *= 1e-3
    elif case == "nonfinite_value":
        v[0, 0, 700, 3] = float("inf")
    elif case == "huge_keys_late":
        k[:, 1::2, 640:] *= 1e5          # these heads switch to fp32 at key 640 ...
        q[:, 1::2] *= 1e-5               # ... with scores that stay of order one there (a well-conditioned softmax)
    elif case == "huge_queries":
        q[:, ::2, 100:200] *= 2e5        # the workgroups holding these queries multiply in fp32 from the start
    elif case == "tiny_keys_large_queries":
        k *= 1e-3                        # low halves and the 2^-11 copies of K deep in the fp16 subnormals
        q *= 300.0
    dout = torch.zeros(B, L, heads * 16)
    out, _, _, _, lse2 = run_core(q, k, v, dout)
    p = torch.softmax(q.double() @ k.double().transpose(-1, -2) / 4.0, dim=-1)
    if case == "nonfinite_value":
        # as in fp32 arithmetic: the column of head 0, image 0 that meets the infinite value is inf (or nan where its
        # probability underflows to 0), everything else is untouched
        col = out[0, :, 3]
        assert not torch.isfinite(col).any()
        v[0, 0, 700, 3] = 0.0
        want = (p @ v.double()).permute(0, 2, 1, 3).reshape(B, L, -1)
        mask = torch.ones_like(out, dtype=torch.bool)
        mask[0, :, 3] = False
        close(out[mask], want[mask], 1e-5, "out beside the non-finite column")
        return
    want = (p @ v.double()).permute(0, 2, 1, 3).reshape(B, L, -1)
    if case.startswith("huge"):
        # per head: the heads without huge values must not be judged against the huge heads' maximum
        for h in range(heads):
            close(out[:, :, 16 * h:16 * h + 16], want[:, :, 16 * h:16 * h + 16], 1e-5, f"out, head {h}")
    else:
        close(out, want, 1e-5, "out")


def test_core_is_deterministic_and_rejects_other_shapes():
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(2, 4, 256, 16, generator=g) for _ in range(3))
    dout = torch.randn(2, 256, 64, generator=g)
    first = run_core(q, k, v, dout)
    second = run_core(q, k, v, dout)
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    lib = hip_engine.load_library()
    x = torch.zeros(4096, device=DEV)
    assert lib.danra_attention_core_fwd(_ptr(x), _ptr(x), _ptr(x), _ptr(x), None, 1, 1, 64, 16, _stream()) == -1   # L % 128
    assert lib.danra_attention_core_fwd(_ptr(x), _ptr(x), _ptr(x), _ptr(x), None, 1, 1, 128, 32, _stream()) == -1  # head_dim
    assert lib.danra_attention_core_bwd(_ptr(x), _ptr(x), _ptr(x), _ptr(x), _ptr(x), None, _ptr(x), _ptr(x), _ptr(x),
                                        _ptr(x), 1, 1, 128, 16, _stream()) == -1                                   # null


@pytest.mark.parametrize("scale", [1.0, 2.5])
def test_core_forward_is_reproducible_on_self_similar_peaked_rows(scale):
    """q = k = v on a coarse grid: every row is dominated by its own key, rows move their softmax reference at
    different tiles and a full-chip grid (B * heads * L / 128 = 512 workgroups) is in flight.  Two launches must agree
    bit for bit and both must be within the core's tolerance of float64 (a form of the kernel that was tried and
    dropped differed from run to run on exactly this data)."""
    B, heads, L = 16, 4, 1024
    g = torch.Generator().manual_seed(2)
    x = torch.randint(-1000, 1001, (B, heads, L, 16), generator=g).float() * (scale / 500.0)
    dout = torch.zeros(B, L, heads * 16)
    first = run_core(x, x, x, dout)[0]
    second = run_core(x, x, x, dout)[0]
    assert torch.equal(first, second)
    xd = x.double()
    want = (torch.softmax(xd @ xd.transpose(-1, -2) / 4.0, dim=-1) @ xd).permute(0, 2, 1, 3).reshape(B, L, -1)
    close(first, want, 1e-5, "out")


def count_calls(monkeypatch, name):
    lib = hip_engine.load_library()
    real = getattr(lib, name)
    calls = []

    def spy(*args):
        calls.append(name)
        return real(*args)
    monkeypatch.setattr(lib, name, spy, raising=False)
    return calls


def test_attention_block_gradients_match_torch_module(monkeypatch):
    """ImageSelfAttention(64, 4) on a 32x32 map: output, input gradient and every parameter gradient against the
    same module evaluated by torch on the CPU (nn.MultiheadAttention's own path)."""
    torch.manual_seed(3)
    block = ImageSelfAttention(64, 4)
    with torch.no_grad():
        block.layernorm.weight.uniform_(0.5, 1.5)
        block.layernorm.bias.uniform_(-0.2, 0.2)
        block.attention.in_proj_bias.uniform_(-0.1, 0.1)
        block.attention.out_proj.bias.uniform_(-0.1, 0.1)
    x = torch.randn(3, 64, 32, 32)
    dy = torch.randn(3, 64, 32, 32)

    xc = x.clone().requires_grad_(True)
    yc = block(xc)
    yc.backward(dy)
    want = {n: p.grad.clone() for n, p in block.named_parameters()}

    gpu = copy.deepcopy(block).to(DEV)
    gpu.zero_grad()
    fwd, bwd = count_calls(monkeypatch, "danra_attention_core_fwd"), count_calls(monkeypatch, "danra_attention_core_bwd")
    xg = x.to(DEV).requires_grad_(True)
    yg = gpu(xg)
    yg.backward(dy.to(DEV))
    assert len(fwd) == 1 and len(bwd) == 1, "the block did not go through the native attention core"
    close(yg.detach().cpu(), yc.detach(), 1e-5, "block output")
    close(xg.grad.cpu(), xc.grad, 1e-5, "input gradient")
    for n, p in gpu.named_parameters():
        close(p.grad.cpu(), want[n], 2e-5, f"gradient of {n}")


@pytest.mark.parametrize("builder", [build_model, build_clean_model])
def test_training_step_gradients(builder, monkeypatch):
    """One training step's loss and parameter gradients (train mode: batch statistics in BatchNorm) on the GPU with the
    64-channel attention cores in the native library, against (a) the same step on the GPU with torch's own attention
    (everything else identical: 5e-3 rms per tensor) and (b) the all-torch evaluation on the CPU (2e-2: torch's GPU and
    CPU convolution backward passes differ by up to 6e-3 on the stem's weight gradients of this two-image batch, with or
    without the native core)."""
    torch.manual_seed(11)
    model = builder()
    model.train()
    B = 2
    x = torch.randn(B, 1, 64, 64)
    t = torch.tensor([17, 640])
    y = torch.tensor([1, 3])
    planes = [torch.randn(B, 1, 64, 64) for _ in range(3)]
    target = torch.randn(B, 1, 64, 64)

    def step(m, dev):
        m.zero_grad()
        pred = m(x.to(dev), t.to(dev), y.to(dev), *(p.to(dev) for p in planes))
        loss = torch.nn.functional.mse_loss(pred, target.to(dev))
        loss.backward()
        return loss.item(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}

    def compare(got, want, tol, what):
        # Root-mean-square comparison per tensor: a pre-activation within rounding of zero may fall on the other side
        # of a ReLU in the two evaluations (seen in layer1 of the clean network), which flips single gradient entries
        # completely while leaving the tensor as a whole nearly unchanged.  A bias in front of an InstanceNorm has no
        # influence on the loss (its computed gradient is rounding noise around zero), hence the floor.
        assert set(got) == set(want)
        floor = 1e-2 * max(w.double().norm().item() / w.numel() ** 0.5 for w in want.values())
        for n in want:
            err = (got[n].double() - want[n].double()).norm().item() / want[n].numel() ** 0.5
            ref = max(want[n].double().norm().item() / want[n].numel() ** 0.5, floor)
            assert err <= tol * ref, f"gradient of {n} against {what}: rms error {err:.3e} against rms {ref:.3e}"

    cpu_loss, cpu_grads = step(model, "cpu")
    gpu = copy.deepcopy(model).to(DEV)
    fwd, bwd = count_calls(monkeypatch, "danra_attention_core_fwd"), count_calls(monkeypatch, "danra_attention_core_bwd")
    loss, grads = step(gpu, DEV)
    assert len(fwd) == 4 and len(bwd) == 4, f"expected the four 64-channel attention blocks on the native core, saw {len(fwd)}/{len(bwd)}"
    monkeypatch.setattr(hip_engine, "attention_core_supported", lambda *a: False)
    torch_loss, torch_grads = step(gpu, DEV)
    assert len(fwd) == 4, "the comparison step must not use the native core"
    assert abs(loss - torch_loss) <= 1e-6 * abs(torch_loss) and abs(loss - cpu_loss) <= 1e-5 * abs(cpu_loss)
    compare(grads, torch_grads, 5e-3, "torch attention on the GPU")
    compare(grads, cpu_grads, 2e-2, "torch on the CPU")


def test_graphed_train_step_equals_the_eager_loop_body():
    """GraphedTrainStep (one captured hipGraph per training step) against the loop body written out
    (reference training_DANRA_conditional.py:608-624), same timesteps and noise: same losses, same parameter changes after
    three SGD steps; and building the object leaves the model as it was."""
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.training import GraphedTrainStep, eager_train_step
    torch.manual_seed(5)
    B = 4
    base = build_model()
    base.train()
    batches = []
    for _ in range(3):
        batches.append(dict(images=torch.randn(B, 1, 64, 64, device=DEV), seasons=torch.randint(0, 4, (B,), device=DEV),
                            cond_images=torch.randn(B, 1, 64, 64, device=DEV), lsm=torch.rand(B, 1, 64, 64, device=DEV),
                            topo=torch.randn(B, 1, 64, 64, device=DEV),
                            t=torch.randint(1, 1000, (B,), device=DEV), noise=torch.randn(B, 1, 64, 64, device=DEV)))
    du = DiffusionUtils(1000, 1e-4, 0.02, device=DEV)
    lossfunc = torch.nn.MSELoss()

    # plain SGD with momentum: the parameter change is then linear in the gradients (Adam's normalised update turns
    # rounding-level differences of near-zero gradient entries into differences of the size of the learning rate)
    eager = copy.deepcopy(base).to(DEV)
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-2, momentum=0.9)
    eager_losses = [eager_train_step(eager, opt_e, du, lossfunc, **b).item() for b in batches]

    graphed = copy.deepcopy(base).to(DEV)
    before = {k: v.detach().clone() for k, v in graphed.state_dict().items()}
    opt_g = torch.optim.SGD(graphed.parameters(), lr=1e-2, momentum=0.9)
    step = GraphedTrainStep(graphed, opt_g, du, lossfunc, batches[0], external_draws=True)
    for k, v in graphed.state_dict().items():
        assert torch.equal(v, before[k]), f"constructing the graphed step changed {k}"
    graphed.eval()
    with torch.no_grad():   # packs the inference weights as they are BEFORE training
        graphed(batches[0]["images"], batches[0]["t"], batches[0]["seasons"], batches[0]["cond_images"], batches[0]["lsm"],
                batches[0]["topo"])
    graphed.train()
    graph_losses = [step(**b).item() for b in batches]

    for a, b in zip(graph_losses, eager_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (graph_losses, eager_losses)
    # what three steps did to every parameter, root-mean-square per tensor (single entries may differ completely where a
    # pre-activation sits within rounding of a ReLU's kink, and torch's convolution weight gradients are not bitwise
    # reproducible between runs)
    se, sg = eager.state_dict(), graphed.state_dict()
    moved = {k: (se[k] - before[k]).double() for k in se if se[k].dtype.is_floating_point and "running" not in k}
    floor = 1e-2 * max(m.norm().item() / m.numel() ** 0.5 for m in moved.values())
    for k, m in moved.items():
        err = ((sg[k] - before[k]).double() - m).norm().item() / m.numel() ** 0.5
        ref = max(m.norm().item() / m.numel() ** 0.5, floor)
        assert err <= 2e-2 * ref, f"{k}: the graphed steps moved it differently (rms {err:.3e} against {ref:.3e})"
    with pytest.raises(RuntimeError):
        step(**{**batches[0], "images": batches[0]["images"][:2]})

    # sampling after graphed steps sees the trained weights (the replay changes them without version bumps)
    graphed.eval()
    with torch.no_grad():
        b0 = batches[0]
        args = (b0["images"], b0["t"], b0["seasons"], b0["cond_images"], b0["lsm"], b0["topo"])
        native = graphed(*args)
        composed = graphed.decoder(*graphed.encoder(args[0], t=args[1], y=args[2], cond_img=args[3], lsm_cond=args[4],
                                                    topo_cond=args[5]), t=args[1])
    close(native.cpu(), composed.cpu(), 1e-4, "inference after graphed training steps")
    graphed.train()

    # draws made inside the graph: every replay sees new timesteps and noise
    opt_i = torch.optim.AdamW(graphed.parameters(), lr=1e-4, capturable=True)
    with pytest.raises(RuntimeError):
        GraphedTrainStep(graphed, torch.optim.AdamW(graphed.parameters(), lr=1e-4), du, lossfunc, batches[0])
    inner = GraphedTrainStep(graphed, opt_i, du, lossfunc, {k: v for k, v in batches[0].items() if k not in ("t", "noise")})
    b0 = {k: v for k, v in batches[0].items() if k not in ("t", "noise")}
    losses = [inner(**b0).item() for _ in range(4)]
    assert all(l == l and l > 0 for l in losses) and len(set(losses)) > 1, losses


@pytest.mark.parametrize("shape,relu", [((4, 64, 16, 16), True), ((3, 128, 8, 8), False), ((2, 512, 2, 2), True),
                                        ((5, 64, 32, 32), False)])
def test_batchnorm_train_kernels_match_torch(shape, relu):
    """danra_batchnorm_train_fwd / _bwd against nn.BatchNorm2d in train mode (+ ReLU) evaluated in float64 on the CPU:
    output, running statistics after two calls, input / weight / bias gradients."""
    torch.manual_seed(21)
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    x1, x2 = torch.randn(shape) * 2 + 0.7, torch.randn(shape) - 0.3
    dy = torch.randn(shape)

    ref = copy.deepcopy(bn).double().train()
    ref(x1.double())
    xr = x2.double().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy.double())

    gpu = copy.deepcopy(bn).to(DEV).train()
    assert hip_engine.batchnorm_train_supported(gpu, x1.to(DEV))
    hip_engine.batchnorm_train(gpu, x1.to(DEV), relu)
    xg = x2.to(DEV).requires_grad_(True)
    yg = hip_engine.batchnorm_train(gpu, xg, relu)
    yg.backward(dy.to(DEV))
    close(yg.detach().cpu(), yr.detach(), 1e-5, "output")
    close(xg.grad.cpu(), xr.grad, 1e-5, "input gradient")
    close(gpu.weight.grad.cpu(), ref.weight.grad, 1e-5, "weight gradient")
    close(gpu.bias.grad.cpu(), ref.bias.grad, 1e-5, "bias gradient")
    close(gpu.running_mean.cpu(), ref.running_mean, 1e-6, "running_mean")
    close(gpu.running_var.cpu(), ref.running_var, 1e-6, "running_var")
    assert int(gpu.num_batches_tracked) == int(ref.num_batches_tracked) == 2


def test_train_mode_network_uses_the_native_batchnorm(monkeypatch):
    model = build_model().to(DEV)
    model.train()
    fwd = count_calls(monkeypatch, "danra_batchnorm_train_fwd")
    bwd = count_calls(monkeypatch, "danra_batchnorm_train_bwd")
    B = 2
    x = torch.randn(B, 1, 64, 64, device=DEV)
    planes = [torch.randn(B, 1, 64, 64, device=DEV) for _ in range(3)]
    model(x, torch.tensor([3, 900], device=DEV), torch.tensor([0, 2], device=DEV), *planes).square().mean().backward()
    # the network has 20 such norms (bn1, 8 blocks x 2, 3 shortcut norms); those whose input arrives channels-last strided
    # (a convolution fed by an attention block's output) stay with torch
    assert len(fwd) == len(bwd) and 10 <= len(fwd) <= 20, (len(fwd), len(bwd))


@pytest.mark.parametrize("shape", [(4, 512, 4, 4), (3, 64, 32, 32), (2, 64, 128, 128), (5, 128, 8, 8)])
def test_instance_norm_kernels_match_torch(shape):
    """danra_instancenorm_fwd / _bwd against nn.InstanceNorm2d evaluated in float64 on the CPU."""
    torch.manual_seed(31)
    norm = torch.nn.InstanceNorm2d(shape[1])
    x = torch.randn(shape) * 3 + 1
    dy = torch.randn(shape)
    xr = x.double().requires_grad_(True)
    yr = norm(xr)
    yr.backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    yg = hip_engine.instance_norm(norm, xg)
    assert yg.grad_fn is not None and "InstanceNorm" in type(yg.grad_fn).__name__
    yg.backward(dy.to(DEV))
    close(yg.detach().cpu(), yr.detach(), 1e-5, "output")
    close(xg.grad.cpu(), xr.grad, 1e-5, "input gradient")
