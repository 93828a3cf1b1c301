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


def run_core_qkv(q, k, v, dout):
    """The same core on the layout of the block's fused QKV projection, [B][L][3][heads*16], read and written in place."""
    lib = hip_engine.load_library()
    B, heads, L, hd = q.shape
    qkv = torch.stack([t.permute(0, 2, 1, 3).reshape(B, L, heads * hd) for t in (q, k, v)], dim=2).to(DEV).contiguous()
    dd = dout.to(DEV).contiguous()
    out = torch.empty((B, L, heads * hd), device=DEV)
    lse2 = torch.empty((B * heads, L), device=DEV)
    delta = torch.empty_like(lse2)
    dqkv = torch.full_like(qkv, float("nan"))
    _check(lib, lib.danra_attention_core_qkv_fwd(_ptr(qkv), _ptr(out), _ptr(lse2), B, heads, L, hd, _stream()), "qkv fwd")
    _check(lib, lib.danra_attention_core_qkv_bwd(_ptr(qkv), _ptr(out), _ptr(dd), _ptr(lse2), _ptr(dqkv), _ptr(delta), B, heads,
                                                 L, hd, _stream()), "qkv bwd")
    torch.cuda.synchronize()
    dq, dk, dv = (dqkv[:, :, i].reshape(B, L, heads, hd).permute(0, 2, 1, 3).contiguous().cpu() for i in range(3))
    return out.cpu(), dq, dk, dv, lse2.cpu()


def run_core_presplit(q, k, v):
    """The same core as the network's inference path runs it: K and V pre-split by their producer."""
    lib = hip_engine.load_library()
    B, heads, L, hd = q.shape
    qd, kd, vd = (t.to(DEV).contiguous() for t in (q, k, v))
    out = torch.empty((B, L, heads * hd), device=DEV)
    _check(lib, lib.danra_attention_core_presplit_fwd(_ptr(qd), _ptr(kd), _ptr(vd), _ptr(out), B, heads, L, hd, _stream()),
           "presplit fwd")
    torch.cuda.synchronize()
    return out.cpu()


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
                                  "nonfinite_value", "huge_keys_late", "huge_queries", "tiny_keys_large_queries",
                                  "band_values_3e-3", "band_all_uniform_2^-8_2^-3"])
@pytest.mark.parametrize("form", ["fp32_operands", "presplit_operands"])
def test_core_forward_value_ranges(case, form):
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
    elif case == "band_values_3e-3":
        v *= 3e-3                        # split path taken, low halves of V small: relative error, not absolute, must hold
    elif case == "band_all_uniform_2^-8_2^-3":
        def band(t):
            return torch.exp2(torch.rand(t.shape, generator=g) * 5.0 - 8.0) * torch.sign(t)
        q, k, v = band(q) * 64.0, band(k), band(v)   # scores of order one from keys in the band
    dout = torch.zeros(B, L, heads * 16)
    if form == "presplit_operands":
        # (here the switch to fp32 products is made per tile of 64 keys, from the flags of its groups of 16, and a flagged
        # group is read from the fp32 tensors while its neighbours come from the halves)
        out = run_core_presplit(q, k, v)
    else:
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


@pytest.mark.parametrize("B,heads,L", [(2, 4, 256), (3, 4, 1024), (1, 2, 4096)])
def test_presplit_core_equals_the_core_on_fp32_operands_bit_for_bit(B, heads, L):
    """With every value inside the split's range the two forms multiply the same halves in the same order."""
    g = torch.Generator().manual_seed(77 + L)
    q = torch.randn(B, heads, L, 16, generator=g)
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g) * 2 + 0.5
    out, _, _, _, _ = run_core(q, k, v, torch.zeros(B, L, heads * 16))
    assert torch.equal(run_core_presplit(q, k, v), out)


def test_presplit_core_reads_single_flagged_groups_from_the_fp32_operands():
    """Groups of 16 keys out of range here and there (K in one place, V in another, the last group of a tile, two in one
    tile), queries in range: each tile with a flagged group is multiplied with fp32 instructions, the flagged groups from the
    fp32 operands and their neighbours from the halves."""
    B, heads, L = 1, 4, 512
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, heads, L, 16, generator=g)
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g)
    v[0, 0, 16:32] *= 2e5           # one V group of tile 0
    v[0, 1, 112:128] *= 1e-6        # a tiny group: last group of tile 1
    k[0, 2, 200, 3] = 4e4           # one K value: group 12 of head 2 (the scores of this key dominate or vanish)
    v[0, 3, 300:310] *= 1e5         # V of groups 18 ...
    k[0, 3, 330:340] *= 5e4         # ... and K of group 20, both in tile 5 (groups 20 and 21), of head 3
    q[0, 3] *= 1e-4
    out = run_core_presplit(q, k, v)
    p = torch.softmax(q.double() @ k.double().transpose(-1, -2) / 4.0, dim=-1)
    want = (p @ v.double()).permute(0, 2, 1, 3).reshape(B, L, -1)
    for h in range(heads):
        close(out[:, :, 16 * h:16 * h + 16], want[:, :, 16 * h:16 * h + 16], 1e-5, f"out, head {h}")


BACKWARD_CASES = ["plain", "tiny_gradient", "huge_gradient", "gradient_rows_of_mixed_size", "zero_gradient_rows",
                  "values_64_and_more", "huge_values_late", "huge_keys_late", "huge_queries", "peaked_rows", "denormal_gradient"]


@pytest.mark.parametrize("case", BACKWARD_CASES)
def test_core_backward_value_ranges(case):
    """The backward kernels multiply on the fp16 matrix cores (operands split in fp16 halves, dO rescaled by a power of two
    per 16 queries in the dQ kernel and per tile of 64 queries in the dK / dV kernel) while their operands allow it, and
    with fp32 instructions otherwise - per wave for its own rows, per tile for the streamed ones.  Every form, and the
    switches between them, within 1e-5 of float64 (relative to the largest entry of each gradient); and the same bits as
    the kernels restricted to fp32 instructions wherever a case forces those anyway."""
    B, heads, L = 2, 4, 512
    g = torch.Generator().manual_seed(321)
    q = torch.randn(B, heads, L, 16, generator=g)
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g)
    dout = torch.randn(B, L, heads * 16, generator=g)
    if case == "tiny_gradient":
        dout *= 3e-9                         # far below the fp16 range before the rescaling
    elif case == "huge_gradient":
        dout *= 7e8
    elif case == "gradient_rows_of_mixed_size":
        dout *= torch.exp2(torch.randint(-30, 10, (B, L, 1), generator=g).float())   # every row its own magnitude
    elif case == "zero_gradient_rows":
        dout[:, 100:300] = 0.0               # whole tiles (and groups of 16 queries) without a gradient
    elif case == "values_64_and_more":
        v *= 150.0                           # 256 max|V| leaves the range a dS may take: fp32 instructions
    elif case == "huge_values_late":
        v[:, 1::2, 320:] *= 3e5              # the key blocks from 320 on (every second head) multiply in fp32
    elif case == "huge_keys_late":
        k[:, 1::2, 320:] *= 1e5
        q[:, 1::2] *= 1e-5
    elif case == "huge_queries":
        q[:, ::2, 100:200] *= 2e5            # beyond the halves (the waves and tiles that hold them multiply in fp32) ...
        k[:, ::2] *= 1e-5                    # ... with scores that stay of order one (a well-conditioned softmax)
    elif case == "peaked_rows":
        q *= 12.0
    elif case == "denormal_gradient":
        dout *= 1e-41                        # cannot be rescaled: fp32 instructions
    want_out, want_dq, want_dk, want_dv = core_reference(q, k, v, dout)
    out, dq, dk, dv, _ = run_core(q, k, v, dout)
    close(out, want_out, 1e-5, "out")
    if case == "denormal_gradient":
        assert all(torch.isfinite(t).all() for t in (dq, dk, dv))
        return
    close(dq, want_dq, 1e-5, "dq")
    close(dk, want_dk, 1e-5, "dk")
    close(dv, want_dv, 1e-5, "dv")
    again = run_core(q, k, v, dout)
    for a, b2, what in zip((dq, dk, dv), again[1:4], ("dq", "dk", "dv")):
        assert torch.equal(a, b2), f"{what}: two calls differ"


def test_core_backward_propagates_nonfinite_gradients():
    """An infinite or NaN entry of dO: the affected gradients are not finite, the others are untouched, nothing hangs."""
    B, heads, L = 1, 2, 256
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, heads, L, 16, generator=g) for _ in range(3))
    dout = torch.randn(B, L, heads * 16, generator=g)
    clean = run_core(q, k, v, dout)
    dout[0, 37, 3] = float("nan")            # head 0, query 37
    _, dq, dk, dv, _ = run_core(q, k, v, dout)
    assert not torch.isfinite(dq[0, 0, 37]).all() and not torch.isfinite(dv[0, 0]).all()
    for got, ref in zip((dq, dk, dv), clean[1:4]):
        assert torch.equal(got[0, 1], ref[0, 1]), "the other head changed"
    keep = torch.ones(L, dtype=torch.bool)
    keep[37] = False
    assert torch.isfinite(dq[0, 0, keep]).all(), "queries other than the poisoned one lost their dq"


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


@pytest.mark.parametrize("B,heads,L,fp32", [(2, 4, 256, 0), (3, 4, 1024, 0), (3, 2, 128, 0), (2, 4, 256, 1)])
def test_core_on_the_projection_layout_equals_the_per_head_layout(B, heads, L, fp32):
    """danra_attention_core_qkv_fwd / _bwd read q, k, v from - and write their gradients into - the [B][L][3][heads*16]
    layout of nn.MultiheadAttention's fused in_proj; the arithmetic is that of danra_attention_core_fwd / _bwd on per-head
    tensors, so every output agrees bit for bit (both product forms), values that leave the fp16 range included."""
    g = torch.Generator().manual_seed(77 + L)
    q = torch.randn(B, heads, L, 16, generator=g) * 3
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g) * 2 + 0.5
    v[0, 1, 40:48] *= 1e6          # a tile of values beyond the split-operand range
    dout = torch.randn(B, L, heads * 16, generator=g)
    lib = hip_engine.load_library()
    lib.danra_set_fp32_products(fp32)
    try:
        want = run_core(q, k, v, dout)
        got = run_core_qkv(q, k, v, dout)
    finally:
        lib.danra_set_fp32_products(0)
    for a, b, what in zip(got, want, ("out", "dq", "dk", "dv", "lse2")):
        assert torch.equal(a, b), f"{what} differs between the two layouts"


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
    fwd, bwd = count_calls(monkeypatch, "danra_attention_core_qkv_fwd"), count_calls(monkeypatch, "danra_attention_core_qkv_bwd")
    xg = x.to(DEV).requires_grad_(True)
    yg = gpu(xg)
    yg.backward(dy.to(DEV))
    assert len(fwd) == 1 and len(bwd) == 1, "the block did not go through the native attention core"
    close(yg.detach().cpu(), yc.detach(), 1e-5, "block output")
    close(xg.grad.cpu(), xc.grad, 1e-5, "input gradient")
    for n, p in gpu.named_parameters():
        close(p.grad.cpu(), want[n], 2e-5, f"gradient of {n}")


def float64_copy(model):
    """A float64 copy of the network for the CPU; the two sinusoidal embeddings stay the float32 expressions every other
    evaluation uses (they are inputs of the arithmetic under test) and are cast on the way out."""
    from diffusionmodelscustom_amd import modules_DANRA_conditional as mods
    m = copy.deepcopy(model).double()
    for mod in m.modules():
        if isinstance(mod, mods.SinusoidalEmbedding):
            mod.forward = (lambda f: (lambda x: f(x).double()))(mod.forward)
    pe = m.encoder.pos_encoding
    m.encoder.pos_encoding = lambda t, ch: pe(t, ch).double()
    return m


def record_relu_decisions(monkeypatch, model):
    """Every ReLU of a GPU evaluation, in program order, as the boolean map `output > 0`: the ones fused into the native
    BatchNorm kernels (seen through the two host functions that launch them) and the nn.ReLU modules.  Returns the
    list the maps are appended to."""
    decisions = []
    real_bn, real_add = hip_engine.batchnorm_train, hip_engine.batchnorm_add_relu_train

    def bn_relu(bn, x, relu=False):
        y = real_bn(bn, x, relu)
        if relu:
            decisions.append((y.detach() > 0).cpu())
        return y

    def bn_add_relu(bn, x, res):
        y = real_add(bn, x, res)
        decisions.append((y.detach() > 0).cpu())
        return y
    monkeypatch.setattr(hip_engine, "batchnorm_train", bn_relu)
    monkeypatch.setattr(hip_engine, "batchnorm_add_relu_train", bn_add_relu)
    for mod in model.modules():
        if isinstance(mod, torch.nn.ReLU):
            mod.register_forward_hook(lambda m, inp, out: decisions.append((out.detach() > 0).cpu()))
    return decisions


def replay_relu_decisions(model, decisions):
    """Makes every nn.ReLU of `model` multiply by the next recorded map instead of deciding for itself; returns the
    iterator (exhausted when the evaluation met exactly the recorded ReLUs)."""
    it = iter(decisions)
    for mod in model.modules():
        if isinstance(mod, torch.nn.ReLU):
            mod.forward = lambda x, it=it: x * next(it).to(x.dtype)
    return it


@pytest.mark.parametrize("builder", [build_model, build_clean_model])
def test_training_step_gradients(builder, monkeypatch):
    """One training step's loss and EVERY parameter tensor's gradient (train mode: batch statistics in BatchNorm)
    against a float64 evaluation of the same step on the CPU, for three float32 evaluations on the GPU: the 64-channel
    attention cores in the native library with their default split-operand fp16 products, the same kernels with fp32
    matrix instructions only, and torch's own attention core.  Bar, per tensor: the rms error at most 1e-5 of the tensor's
    rms (measured: 1.8e-6 to 2.8e-6 in the worst tensor of every evaluation); loss within 2e-6.

    The gradient of this piecewise-linear network is discontinuous wherever a ReLU input crosses zero, and of its few
    million ReLU inputs a handful always sit within float32 rounding of zero, so two correct evaluations can land on
    different sides (measured, tools/diag_train_grad.py: one such flip in encoder.layer1 moves the stem's weight gradient
    by 3.8e-3 of its rms, and which side a float32 evaluation lands on changes from run to run because the step contains
    kernels that accumulate in no fixed order).  A bar that has to absorb flips cannot be tight, so the flips are taken out
    of the comparison instead: each GPU evaluation records the decision of every ReLU element it evaluates (about 4
    million booleans, in program order), and the float64 evaluation it is compared with multiplies by those recorded maps
    instead of deciding for itself.  Both then differentiate the same smooth branch of the network, and what remains is
    the arithmetic of the kernels.  (Wherever the float64 pre-activation has the recorded sign - all but the handful of
    elements above - the replayed map IS the ReLU.)"""
    torch.manual_seed(11)
    model = builder()
    model.train()
    B = 8
    x = torch.randn(B, 1, 64, 64)
    t = torch.tensor([17, 640, 3, 999, 250, 501, 64, 820])
    y = torch.tensor([1, 3, 0, 2, 2, 1, 3, 0])
    planes = [torch.randn(B, 1, 64, 64) for _ in range(3)]
    target = torch.randn(B, 1, 64, 64)

    def step(m, dev, dtype=torch.float32):
        m.zero_grad()
        pred = m(x.to(dev, dtype), t.to(dev), y.to(dev), *(p.to(dev, dtype) for p in planes))
        loss = torch.nn.functional.mse_loss(pred, target.to(dev, dtype))
        loss.backward()
        return loss.item(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}

    reference = float64_copy(model)

    def against_float64(loss, got, decisions):
        """(relative loss error, the worst tensor's relative rms error, its name) of one GPU evaluation against the
        float64 evaluation that replays its ReLU decisions."""
        left = replay_relu_decisions(reference, decisions)
        want_loss, want = step(reference, "cpu", torch.float64)
        assert next(left, None) is None, "the float64 evaluation met fewer ReLUs than the GPU evaluation recorded"
        assert set(got) == set(want)
        # A bias in front of an InstanceNorm has no influence on the loss (its gradient is rounding noise around
        # zero), hence the floor on the reference rms.
        rms = {n: w.norm().item() / w.numel() ** 0.5 for n, w in want.items()}
        floor = 1e-2 * max(rms.values())
        rel = {n: (got[n] - want[n]).norm().item() / want[n].numel() ** 0.5 / max(rms[n], floor) for n in want}
        n = max(rel, key=rel.get)
        return abs(loss - want_loss) / abs(want_loss), rel[n], n

    lib = hip_engine.load_library()
    gpu = copy.deepcopy(model).to(DEV)
    fwd, bwd = count_calls(monkeypatch, "danra_attention_core_qkv_fwd"), count_calls(monkeypatch, "danra_attention_core_qkv_bwd")
    decisions = record_relu_decisions(monkeypatch, gpu)
    report = {}
    try:
        for name, fp32_products in (("native core, split-operand fp16 products", 0), ("native core, fp32 products", 1)):
            lib.danra_set_fp32_products(fp32_products)
            del fwd[:], bwd[:], decisions[:]
            loss, grads = step(gpu, DEV)
            assert len(fwd) == 4 and len(bwd) == 4, \
                f"expected the four 64-channel attention blocks on the native core, saw {len(fwd)}/{len(bwd)}"
            report[name] = against_float64(loss, grads, list(decisions))
    finally:
        lib.danra_set_fp32_products(0)
    monkeypatch.setattr(hip_engine, "attention_core_supported", lambda *a: False)
    del fwd[:], decisions[:]
    loss, grads = step(gpu, DEV)
    assert len(fwd) == 0, "the comparison step must not use the native core"
    report["torch attention on the GPU"] = against_float64(loss, grads, list(decisions))
    text = "; ".join(f"{k}: loss {v[0]:.1e}, worst gradient {v[1]:.2e} ({v[2]})" for k, v in report.items())
    print(text)
    for name, (loss_err, err, tensor) in report.items():
        assert loss_err <= 2e-6, f"{name}: loss off by {loss_err:.2e} relative to float64 [{text}]"
        assert err <= 1e-5, f"{name}: gradient of {tensor} is {err:.3e} of its rms away from float64 [{text}]"


def test_graphed_train_step_equals_the_eager_loop_body():
    """GraphedTrainStep (one captured hipGraph per training step) against the loop body written out
    (reference training_DANRA_conditional.py:608-624), same timesteps and noise; and building the object leaves the model
    as it was.

    The first step starts from identical parameters, so its forward pass - and with it every ReLU mask - is the same
    computation in both forms and only the summation order of torch's convolution weight gradients (atomics) can differ:
    the parameter changes of that step agree to 1e-3 of their rms.  From the second step on the parameters differ in the
    last bits, a ReLU input within rounding of zero may then fall on the other side in one of the two runs, and a single
    such flip moves a tensor's gradient by up to a percent of its rms (see test_training_step_gradients): after three
    steps the losses agree to 1e-4 and the accumulated parameter changes to 5e-2 of their rms."""
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.training import GraphedTrainStep, eager_train_step
    torch.manual_seed(5)
    B = 8
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

    def trained(sd):
        return {k: v.detach().clone() for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}

    # plain SGD with momentum: the parameter change is then linear in the gradients (Adam's normalised update turns
    # rounding-level differences of near-zero gradient entries into differences of the size of the learning rate)
    eager = copy.deepcopy(base).to(DEV)
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-2, momentum=0.9)
    eager_losses, eager_after = [], []
    for b in batches:
        eager_losses.append(eager_train_step(eager, opt_e, du, lossfunc, **b).item())
        eager_after.append(trained(eager.state_dict()))

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
    graph_losses, graph_after = [], []
    for b in batches:
        graph_losses.append(step(**b).item())
        graph_after.append(trained(graphed.state_dict()))

    for a, b in zip(graph_losses, eager_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (graph_losses, eager_losses)
    # what the steps did to every parameter, root-mean-square per tensor
    for n_steps, tol in ((1, 1e-3), (3, 5e-2)):
        moved = {k: (v - before[k]).double() for k, v in eager_after[n_steps - 1].items()}
        floor = 1e-2 * max(m.norm().item() / m.numel() ** 0.5 for m in moved.values())
        for k, m in moved.items():
            err = ((graph_after[n_steps - 1][k] - before[k]).double() - m).norm().item() / m.numel() ** 0.5
            ref = max(m.norm().item() / m.numel() ** 0.5, floor)
            assert err <= tol * ref, \
                f"{k}: {n_steps} graphed step(s) moved it differently (rms {err:.3e} against {ref:.3e}, bar {tol:g})"
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


@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (3, 128, 8, 8), (5, 512, 2, 2)])
def test_batchnorm_add_relu_kernels_match_torch(shape):
    """danra_batchnorm_train_add_relu_fwd / _bwd - the tail of a BasicBlock, relu(bn2(z) + shortcut) - against the torch
    composition in float64 on the CPU: output, running statistics, gradients of both inputs and of the norm's parameters."""
    torch.manual_seed(22)
    bn = torch.nn.BatchNorm2d(shape[1])
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x, res, dy = torch.randn(shape) * 2 + 0.7, torch.randn(shape), torch.randn(shape)
    ref = copy.deepcopy(bn).double().train()
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    yr = torch.relu(ref(xr) + rr)
    yr.backward(dy.double())
    gpu = copy.deepcopy(bn).to(DEV).train()
    xg, rg = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    yg = hip_engine.batchnorm_add_relu_train(gpu, xg, rg)
    yg.backward(dy.to(DEV))
    close(yg.detach().cpu(), yr.detach(), 1e-5, "output")
    close(xg.grad.cpu(), xr.grad, 1e-5, "input gradient")
    close(rg.grad.cpu(), rr.grad, 1e-5, "shortcut gradient")
    close(gpu.weight.grad.cpu(), ref.weight.grad, 1e-5, "weight gradient")
    close(gpu.bias.grad.cpu(), ref.bias.grad, 1e-5, "bias gradient")
    close(gpu.running_mean.cpu(), ref.running_mean, 1e-6, "running_mean")
    close(gpu.running_var.cpu(), ref.running_var, 1e-6, "running_var")
    assert int(gpu.num_batches_tracked) == 1


@pytest.mark.parametrize("M,N", [(32768, 192), (8192, 64), (4096 + 6, 128), (131072, 192), (70, 64), (1, 192)])
def test_linear_weight_gradient_kernel_against_float64(M, N):
    """danra_linear_wgrad (dW = dy^T x, db = column sums of dy over M token rows) against a float64 evaluation: the
    attention projections' shapes at batch 32 (1024 and 256 tokens) and batch 32 at 4096 tokens, a row count that is
    neither even per slice nor a multiple of the slice, and fewer rows than one slice.  Per-op bar 1e-5 of the largest
    value; two calls give the same bits."""
    K = 64
    lib = hip_engine.load_library()
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, N, generator=g)
    x = torch.randn(M, K, generator=g) + 0.5
    want_w = dy.double().t() @ x.double()
    want_b = dy.double().sum(0)
    dyd, xd = dy.to(DEV), x.to(DEV)
    assert lib.danra_linear_wgrad_supported(M, N, K) == 1
    ws = torch.empty(lib.danra_linear_wgrad_workspace(M, N, K) // 4, device=DEV)
    outs = []
    for _ in range(2):
        dw = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        rc = lib.danra_linear_wgrad(ctypes.c_void_p(dyd.data_ptr()), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(dw.data_ptr()),
                                    ctypes.c_void_p(db.data_ptr()), M, N, K, ctypes.c_void_p(ws.data_ptr()), None)
        assert rc == 0, lib.danra_last_error()
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(outs[0][0], want_w, 1e-5, "dW")
    close(outs[0][1], want_b, 1e-5, "db")
    # without a bias gradient; refusals
    dw = torch.empty(N, K, device=DEV)
    assert lib.danra_linear_wgrad(ctypes.c_void_p(dyd.data_ptr()), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(dw.data_ptr()),
                                  None, M, N, K, ctypes.c_void_p(ws.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), outs[0][0])
    assert lib.danra_linear_wgrad_supported(M, N, 128) == 0 and lib.danra_linear_wgrad_supported(M, 96, K) == 0
    assert lib.danra_linear_wgrad(ctypes.c_void_p(dyd.data_ptr()), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(dw.data_ptr()),
                                  None, M, 96, K, ctypes.c_void_p(ws.data_ptr()), None) == -1
    assert lib.danra_linear_wgrad(None, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(dw.data_ptr()), None, M, N, K,
                                  ctypes.c_void_p(ws.data_ptr()), None) == -1


def test_token_linear_equals_torch_linear():
    """hip_engine.token_linear against torch.nn.functional.linear evaluated in float64: output and the three gradients
    within 1e-5 (product and input gradient from danra_linear, weight and bias gradients from danra_linear_wgrad), and
    bit-identical between two calls."""
    torch.manual_seed(5)
    x = torch.randn(4, 1024, 64, device=DEV, requires_grad=True)
    w = torch.randn(192, 64, device=DEV, requires_grad=True)
    b = torch.randn(192, device=DEV, requires_grad=True)
    up = torch.randn(4, 1024, 192, device=DEV)
    x64, w64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, w, b))
    y0 = torch.nn.functional.linear(x64, w64, b64)
    g0 = torch.autograd.grad(y0, (x64, w64, b64), up.double().cpu())
    y1 = hip_engine.token_linear(x, w, b)
    g1 = torch.autograd.grad(y1, (x, w, b), up)
    close(y1.detach().cpu(), y0.detach(), 1e-5, "y")
    for a, c, name in zip(g1, g0, ("dx", "dw", "db")):
        close(a.cpu(), c, 1e-5, name)
    y2 = hip_engine.token_linear(x, w, b)
    g2 = torch.autograd.grad(y2, (x, w, b), up)
    assert torch.equal(y1, y2) and all(torch.equal(a, c) for a, c in zip(g1, g2))


@pytest.mark.parametrize("B,C,H,W", [(3, 64, 32, 32), (2, 64, 16, 16), (3, 128, 8, 8), (5, 256, 4, 4), (7, 512, 2, 2),
                                     (2, 64, 64, 64)])
def test_attention_block_entry_and_exit_match_float64(B, C, H, W):
    """hip_engine.ln_tokens (map -> LayerNorm'ed tokens) and hip_engine.tokens_residual_nchw (tokens + residual -> map)
    against the reference's composition (modules_DANRA_conditional.py:93-110: reshape, permute, nn.LayerNorm, add, permute
    back) in float64: outputs and every gradient within 1e-5, bit-identical between two calls."""
    torch.manual_seed(B + C)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_()
    x = (torch.randn(B, C, H, W, device=DEV) * 3 + 0.5).requires_grad_(True)
    tok = torch.randn(B, H * W, C, device=DEV, requires_grad=True)
    up_n, up_o = torch.randn(B, H * W, C, device=DEV), torch.randn(B, C, H, W, device=DEV)
    assert hip_engine.attention_block_supported(ln, x)

    def native():
        normed = hip_engine.ln_tokens(ln, x)
        out = hip_engine.tokens_residual_nchw(tok, x)
        grads = torch.autograd.grad([normed, out], [x, ln.weight, ln.bias, tok], [up_n, up_o])
        return (normed.detach(), out.detach()) + grads

    got = native()
    ln64 = torch.nn.LayerNorm(C).double()
    ln64.load_state_dict({k: v.double().cpu() for k, v in ln.state_dict().items()})
    x64, tok64 = x.detach().double().cpu().requires_grad_(True), tok.detach().double().cpu().requires_grad_(True)
    tokens = x64.reshape(-1, C, H * W).permute(0, 2, 1)
    normed64 = ln64(tokens)
    out64 = (tok64 + tokens).permute(0, 2, 1).reshape(-1, C, H, W)
    want = (normed64.detach(), out64.detach()) + torch.autograd.grad([normed64, out64], [x64, ln64.weight, ln64.bias, tok64],
                                                                      [up_n.double().cpu(), up_o.double().cpu()])
    for a, b, what in zip(got, want, ("normed", "out", "dx", "dgamma", "dbeta", "dtokens")):
        close(a.cpu(), b, 1e-5, what)
    for a, b in zip(got, native()):
        assert torch.equal(a, b)


def native_call_counters(monkeypatch):
    names = ("danra_attention_core_qkv_fwd", "danra_attention_core_qkv_bwd", "danra_batchnorm_train_fwd", "danra_batchnorm_train_bwd",
             "danra_batchnorm_train_add_relu_fwd", "danra_batchnorm_train_add_relu_bwd",
             "danra_batchnorm_train_add_relu_fwd", "danra_batchnorm_train_add_relu_bwd", "danra_instancenorm_fwd", "danra_instancenorm_bwd", "danra_linear_wgrad", "danra_conv3x3_train_fwd",
             "danra_conv3x3_train_dgrad", "danra_conv3x3_train_wgrad", "danra_pointwise_train_fwd",
             "danra_pointwise_train_dgrad", "danra_pointwise_train_wgrad", "danra_ln_tokens_fwd", "danra_ln_tokens_bwd",
             "danra_tokens_to_nchw_add", "danra_attention_small_qkv_fwd", "danra_attention_small_qkv_bwd",
             "danra_conv3x3_c1_train_fwd", "danra_conv3x3_c1_train_dgrad", "danra_conv3x3_c1_train_wgrad",
             "danra_conv8x8s2_train_fwd", "danra_conv8x8s2_train_dgrad", "danra_conv8x8s2_train_wgrad",
             "danra_conv3x3s2_train_fwd", "danra_conv3x3s2_train_dgrad", "danra_conv3x3s2_train_wgrad")
    return {n: count_calls(monkeypatch, n) for n in names}


# the conditional network: 4 attention blocks at 64 channels (core forward and backward, and the weight gradients of their
# two projections); bn1 + 8 blocks x 2 + 3 shortcut norms, the second norm of every block fused with the shortcut add and the ReLU; 4 decoder blocks x 2 InstanceNorms + the final block's first
# (its second is an Identity); the 3x3 convolutions: 4 + 4 + 4 + 4 in the BasicBlocks (the first convolution of layers 2-4
# has stride 2 and goes through the stride-1 kernels) and the four decoder blocks' (the final block's has one output
# channel and kernels of its own), each in three directions; the cores of the five attention blocks at 128 channels and more
EXPECTED_NATIVE_CALLS = {"danra_conv3x3_c1_train_fwd": 1, "danra_conv3x3_c1_train_dgrad": 1, "danra_conv3x3_c1_train_wgrad": 1,
                         "danra_attention_core_qkv_fwd": 4, "danra_attention_core_qkv_bwd": 4, "danra_attention_small_qkv_fwd": 5,
                         "danra_attention_small_qkv_bwd": 5, "danra_batchnorm_train_fwd": 12,
                         "danra_batchnorm_train_bwd": 12, "danra_batchnorm_train_add_relu_fwd": 8,
                         "danra_batchnorm_train_add_relu_bwd": 8, "danra_instancenorm_fwd": 9, "danra_instancenorm_bwd": 9,
                         "danra_linear_wgrad": 8, "danra_conv3x3_train_fwd": 17, "danra_conv3x3_train_dgrad": 17,
                         "danra_conv3x3_train_wgrad": 17,
                         "danra_conv3x3s2_train_fwd": 3, "danra_conv3x3s2_train_dgrad": 3, "danra_conv3x3s2_train_wgrad": 3,
                         # the two 8x8 stride-2 stem convolutions (conv1's input planes ask for no gradient)
                         "danra_conv8x8s2_train_fwd": 2, "danra_conv8x8s2_train_dgrad": 1, "danra_conv8x8s2_train_wgrad": 2,
                         # the three 1x1 stride-2 shortcut convolutions and the five transposed convolutions
                         "danra_pointwise_train_fwd": 8, "danra_pointwise_train_dgrad": 8, "danra_pointwise_train_wgrad": 8,
                         # entry and exit of the nine attention blocks
                         "danra_ln_tokens_fwd": 9, "danra_ln_tokens_bwd": 9, "danra_tokens_to_nchw_add": 9}


def test_train_mode_network_uses_the_native_kernels(monkeypatch):
    """Every train-mode BatchNorm (also the seven whose input arrives with channels-last strides from a convolution fed
    by an attention block), every InstanceNorm and the four 64-channel attention cores of one step run in the library."""
    model = build_model().to(DEV)
    model.train()
    calls = native_call_counters(monkeypatch)
    B = 2
    x = torch.randn(B, 1, 64, 64, device=DEV)
    planes = [torch.randn(B, 1, 64, 64, device=DEV) for _ in range(3)]
    model(x, torch.tensor([3, 900], device=DEV), torch.tensor([0, 2], device=DEV), *planes).square().mean().backward()
    assert {n: len(c) for n, c in calls.items()} == EXPECTED_NATIVE_CALLS


def test_training_step_under_autocast_uses_the_native_kernels(monkeypatch):
    """The reference's training loop runs its forward inside autocast() with a GradScaler whenever CUDA is available
    (training_DANRA_conditional.py:781,795-798,962-984).  Inside that context the native kernels still run (their
    inputs arrive in half precision and are taken as float32), the step trains, and its loss and gradients stay close to
    the float32 step's (half-precision convolutions and linears around them: 2e-2 rms on the large tensors)."""
    torch.manual_seed(13)
    model = build_model().to(DEV)
    model.train()
    B = 4
    x = torch.randn(B, 1, 64, 64, device=DEV)
    t = torch.tensor([3, 900, 411, 77], device=DEV)
    y = torch.tensor([0, 2, 1, 3], device=DEV)
    planes = [torch.randn(B, 1, 64, 64, device=DEV) for _ in range(3)]
    target = torch.randn(B, 1, 64, 64, device=DEV)

    model.zero_grad()
    loss32 = torch.nn.functional.mse_loss(model(x, t, y, *planes), target)
    loss32.backward()
    grads32 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    calls = native_call_counters(monkeypatch)
    scaler = torch.amp.GradScaler("cuda")
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    model.zero_grad()
    with torch.autocast("cuda"):
        pred = model(x, t, y, *planes)
        loss16 = torch.nn.functional.mse_loss(pred, target)
    scaler.scale(loss16).backward()
    scaler.unscale_(opt)
    assert {n: len(c) for n, c in calls.items()} == EXPECTED_NATIVE_CALLS
    assert abs(loss16.item() - loss32.item()) <= 1e-2 * abs(loss32.item()), (loss16.item(), loss32.item())
    floor = 1e-1 * max(g.double().norm().item() / g.numel() ** 0.5 for g in grads32.values())
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        assert torch.isfinite(p.grad).all(), n
        err = (p.grad.double() - grads32[n].double()).norm().item() / p.numel() ** 0.5
        ref = max(grads32[n].double().norm().item() / p.numel() ** 0.5, floor)
        assert err <= 5e-2 * ref, f"{n}: autocast gradient rms error {err:.3e} against {ref:.3e}"


@pytest.mark.parametrize("size,B", [(96, 2), (128, 2)])
def test_training_step_on_other_map_sizes(size, B):
    """One training step on map sizes other than 64x64: 96 (sides that are not powers of two: the convolution kernels and
    some of the attention-block kernels do not take them and torch's operators step in) and 128 (the LUMI configuration).
    32x32 is left out: its deepest map is one pixel, whose instance norm is 0/sqrt(eps) and turns rounding into gradient
    noise on either path (median 2.5e-3 between two float32 evaluations).  The loss and every gradient of the step with the native kernels agree with the same
    step on torch's own operators (2e-6 on the loss; per tensor the rms difference at most 2e-2 of the tensor's rms, with the
    median over the tensors at most 1e-3: float32 against float32, flips of ReLU masks included)."""
    torch.manual_seed(size)
    model = build_model(size=size).to(DEV)
    model.train()
    x = torch.randn(B, 1, size, size, device=DEV)
    t = torch.randint(1, 1000, (B,), device=DEV)
    y = torch.randint(0, 4, (B,), device=DEV)
    planes = [torch.randn(B, 1, size, size, device=DEV) for _ in range(3)]
    target = torch.randn(B, 1, size, size, device=DEV)

    def step():
        model.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x, t, y, *planes), target)
        loss.backward()
        return loss.item(), {n: p.grad.detach().double() for n, p in model.named_parameters() if p.grad is not None}

    running = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    loss_n, got = step()
    model.load_state_dict(running, strict=False)  # the same running statistics for the second evaluation
    hip_engine.native_training_kernels = False
    try:
        loss_t, want = step()
    finally:
        hip_engine.native_training_kernels = True
    assert abs(loss_n - loss_t) <= 2e-6 * abs(loss_t), (loss_n, loss_t)
    assert set(got) == set(want)
    rms = {n: w.norm().item() / w.numel() ** 0.5 for n, w in want.items()}
    floor = 1e-2 * max(rms.values())
    rel = {n: (got[n] - want[n]).norm().item() / want[n].numel() ** 0.5 / max(rms[n], floor) for n in want}
    worst = max(rel, key=rel.get)
    median = sorted(rel.values())[len(rel) // 2]
    print(f"size {size}: loss {loss_n:.6f} / {loss_t:.6f}, worst {worst} {rel[worst]:.2e}, median {median:.2e}")
    assert rel[worst] <= 2e-2, f"{worst}: {rel[worst]:.3e}"
    assert median <= 1e-3


def test_graphed_step_leaves_a_resumed_optimizer_as_it_was():
    """Constructing GraphedTrainStep on an optimizer that already holds moments and step counts (a run resumed with
    load_state_dict) changes neither them nor the model nor the random generators' states."""
    from diffusionmodelscustom_amd import DiffusionUtils
    from diffusionmodelscustom_amd.training import GraphedTrainStep, eager_train_step
    torch.manual_seed(6)
    B = 2
    model = build_model().to(DEV)
    model.train()
    batch = dict(images=torch.randn(B, 1, 64, 64, device=DEV), seasons=torch.randint(0, 4, (B,), device=DEV),
                 cond_images=torch.randn(B, 1, 64, 64, device=DEV), lsm=torch.rand(B, 1, 64, 64, device=DEV),
                 topo=torch.randn(B, 1, 64, 64, device=DEV))
    du = DiffusionUtils(1000, 1e-4, 0.02, device=DEV)
    lossfunc = torch.nn.MSELoss()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True)
    for _ in range(2):
        eager_train_step(model, opt, du, lossfunc, **batch)
    state_before = copy.deepcopy(opt.state_dict()["state"])
    model_before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert any(float(st["exp_avg"].abs().max()) > 0 for st in state_before.values())
    cpu_rng, dev_rng = torch.get_rng_state(), torch.cuda.get_rng_state(DEV)
    step = GraphedTrainStep(model, opt, du, lossfunc, batch)
    assert torch.equal(torch.get_rng_state(), cpu_rng) and torch.equal(torch.cuda.get_rng_state(DEV), dev_rng)
    state_after = opt.state_dict()["state"]
    assert set(state_after) == set(state_before)
    for idx, st in state_before.items():
        for k, v in st.items():
            same = torch.equal(state_after[idx][k], v) if torch.is_tensor(v) else state_after[idx][k] == v
            assert same, f"optimizer state {k} of parameter {idx} was changed by the warm-up"
    for k, v in model.state_dict().items():
        assert torch.equal(v, model_before[k]), k
    assert float(step(**batch).item()) > 0
    assert int(next(iter(opt.state_dict()["state"].values()))["step"]) == 3  # two eager steps + one replay

    sgd = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, dampening=0.1)
    with pytest.raises(RuntimeError, match="dampening"):
        GraphedTrainStep(model, sgd, du, lossfunc, batch)


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


def small_reference(q, k, v, dout):
    """softmax(q k^T / sqrt(hd)) v and its gradients in float64 (what nn.MultiheadAttention evaluates between its
    projections, modules_DANRA_conditional.py:100-107), out in token order [B, L, heads*hd]."""
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    B, heads, L, hd = q.shape
    p = torch.softmax(qd @ kd.transpose(-1, -2) / hd ** 0.5, dim=-1)
    out = (p @ vd).permute(0, 2, 1, 3).reshape(B, L, heads * hd)
    out.backward(dout.double())
    return out.detach(), qd.grad, kd.grad, vd.grad


@pytest.mark.parametrize("B,heads,L,hd,qscale", [
    (3, 4, 64, 32, 1.0),     # the 8x8 level of a 64x64 input
    (5, 4, 16, 64, 1.0),     # 4x4
    (7, 4, 4, 128, 1.0),     # 2x2
    (2, 2, 64, 32, 6.0),     # peaked rows
    (2, 1, 16, 128, 1.0),    # one head of 128
    (1, 4, 1, 64, 1.0),      # a single token: every probability is one
    (32, 4, 64, 32, 1.0),    # a training batch
])
def test_short_sequence_core_forward_and_backward_match_float64(B, heads, L, hd, qscale):
    lib = hip_engine.load_library()
    assert lib.danra_attention_small_supported(B, heads, L, hd)
    g = torch.Generator().manual_seed(L * 100 + hd)
    q = torch.randn(B, heads, L, hd, generator=g) * qscale
    k = torch.randn(B, heads, L, hd, generator=g)
    v = torch.randn(B, heads, L, hd, generator=g) * 2 + 0.5
    dout = torch.randn(B, L, heads * hd, generator=g)
    qd, kd, vd, dd = (t.to(DEV).contiguous() for t in (q, k, v, dout))
    out = torch.empty((B, L, heads * hd), device=DEV)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    _check(lib, lib.danra_attention_small_fwd(_ptr(qd), _ptr(kd), _ptr(vd), _ptr(out), B, heads, L, hd, _stream()), "fwd")
    _check(lib, lib.danra_attention_small_bwd(_ptr(qd), _ptr(kd), _ptr(vd), _ptr(dd), _ptr(dq), _ptr(dk), _ptr(dv), B, heads, L, hd,
                                              _stream()), "bwd")
    torch.cuda.synchronize()
    want_out, want_dq, want_dk, want_dv = small_reference(q, k, v, dout)
    close(out.cpu(), want_out, 1e-5, "out")
    close(dv.cpu(), want_dv, 1e-5, "dv")
    close(dq.cpu(), want_dq, 1e-5, "dq")
    close(dk.cpu(), want_dk, 1e-5, "dk")
    # the same kernels on the layout of the fused QKV projection, [B][L][3][heads*hd], read and written in place: bit for bit
    qkv = torch.stack([t.permute(0, 2, 1, 3).reshape(B, L, heads * hd) for t in (q, k, v)], dim=2).to(DEV).contiguous()
    out2, dqkv = torch.empty_like(out), torch.full_like(qkv, float("nan"))
    _check(lib, lib.danra_attention_small_qkv_fwd(_ptr(qkv), _ptr(out2), B, heads, L, hd, _stream()), "qkv fwd")
    _check(lib, lib.danra_attention_small_qkv_bwd(_ptr(qkv), _ptr(dd), _ptr(dqkv), B, heads, L, hd, _stream()), "qkv bwd")
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    for i, want in enumerate((dq, dk, dv)):
        assert torch.equal(dqkv[:, :, i].reshape(B, L, heads, hd).permute(0, 2, 1, 3), want), "qkv"[i]


def test_short_sequence_core_refuses_other_shapes_and_runs_in_the_attention_block():
    lib = hip_engine.load_library()
    for B, heads, L, hd in [(2, 4, 128, 16), (2, 4, 64, 64), (2, 4, 16, 130), (0, 4, 16, 64), (2, 4, 16, 6)]:
        assert not lib.danra_attention_small_supported(B, heads, L, hd), (B, heads, L, hd)
    x = torch.zeros(16, device=DEV)
    rc = lib.danra_attention_small_fwd(_ptr(x), _ptr(x), _ptr(x), _ptr(x), 2, 4, 128, 16, _stream())
    assert rc != 0 and b"short-sequence core" in lib.danra_last_error()
    # through the module hook: nn.MultiheadAttention of a deep level against torch's own evaluation of the module
    torch.manual_seed(3)
    mha = torch.nn.MultiheadAttention(128, 4, batch_first=True).to(DEV)
    tokens = torch.randn(6, 64, 128, device=DEV, requires_grad=True)
    calls = []
    real = hip_engine._AttentionCoreSmallQkv.apply
    try:
        hip_engine._AttentionCoreSmallQkv.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        got = hip_engine.differentiable_self_attention(mha, tokens)
    finally:
        hip_engine._AttentionCoreSmallQkv.apply = real
    assert calls, "the native core did not run"
    g = torch.randn_like(got)
    got.backward(g)
    got_grads = [tokens.grad.clone()] + [p.grad.clone() for p in mha.parameters()]
    tokens.grad = None
    mha.zero_grad()
    want = mha(tokens, tokens, tokens, need_weights=False)[0]
    want.backward(g)
    want_grads = [tokens.grad] + [p.grad for p in mha.parameters()]
    close(got.detach().cpu(), want.detach().cpu(), 2e-5, "block output")
    for a, b in zip(got_grads, want_grads):
        close(a.cpu(), b.cpu(), 5e-5, "gradient")
