"""ctypes binding of libdanra_hip.so (C ABI: include/danra_hip.h).

PyTorch supplies device memory and streams only; every kernel lives in the shared library.
Loading fails loudly: there is no software fallback for the HIP path.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdanra_hip.so")
_lib = None
_lib_lock = threading.Lock()

NOISE_HOST = 0
NOISE_PHILOX = 1

# The kernels of the differentiable path (attention core, train-mode BatchNorm, InstanceNorm: forward and backward).
# False sends that path through torch's own operators everywhere - for comparisons; the inference path is not affected.
native_training_kernels = True
# the 3x3 stride-1 convolutions of the training step in the native library as well (False: torch's own, for comparisons)
native_training_convolutions = True


class DanraConfig(ctypes.Structure):
    _fields_ = [
        ("has_lsm", ctypes.c_int),
        ("has_topo", ctypes.c_int),
        ("has_cond", ctypes.c_int),
        ("num_classes", ctypes.c_int),
        ("time_embedding", ctypes.c_int),
        ("out_channels", ctypes.c_int),
        ("n_heads", ctypes.c_int),
        ("block_layers", ctypes.c_int * 4),
        ("attn_ff", ctypes.c_int),
    ]


_P = ctypes.c_void_p
_SIGNATURES = {
    "danra_last_error": (ctypes.c_char_p, []),
    "danra_create": (ctypes.c_int, [ctypes.POINTER(DanraConfig), ctypes.POINTER(_P)]),
    "danra_destroy": (ctypes.c_int, [_P]),
    "danra_num_params": (ctypes.c_int, [_P]),
    "danra_param_name": (ctypes.c_char_p, [_P, ctypes.c_int]),
    "danra_param_numel": (ctypes.c_size_t, [_P, ctypes.c_int]),
    "danra_set_param": (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_size_t]),
    "danra_pack_weights": (ctypes.c_int, [_P, _P]),
    "danra_set_freq_tables": (ctypes.c_int, [_P, _P, _P, ctypes.c_int]),
    "danra_unet_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_get_fmaps": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "danra_ddpm_update": (ctypes.c_int, [_P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_size_t, _P]),
    "danra_ddpm_update_philox": (ctypes.c_int, [_P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_size_t,
                                                ctypes.c_uint64, ctypes.c_uint64, _P]),
    "danra_sample": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    _P, _P, _P, ctypes.c_int, _P, ctypes.c_uint64, _P, ctypes.c_int, _P, _P]),
    "danra_sample_guided": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           _P, _P, _P, ctypes.c_int, _P, ctypes.c_uint64, ctypes.c_float, ctypes.c_float,
                                           _P, ctypes.c_int, _P, _P]),
    "danra_noise_image": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                                         ctypes.c_uint64, ctypes.c_uint64, _P]),
    "danra_attention_core_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_attention_core_presplit_fwd": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_set_fp32_products": (ctypes.c_int, [ctypes.c_int]),
    "danra_conv3x3s2_train_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "danra_conv3x3s2_train_workspace": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "danra_conv3x3s2_train_fwd": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3s2_train_dgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3s2_train_wgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv8x8s2_train_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "danra_conv8x8s2_train_dgrad_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "danra_conv8x8s2_train_workspace": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "danra_conv8x8s2_train_fwd": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P]),
    "danra_conv8x8s2_train_dgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P]),
    "danra_conv8x8s2_train_wgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3_c1_train_supported": (ctypes.c_int, [ctypes.c_int] * 4),
    "danra_conv3x3_c1_train_workspace": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "danra_conv3x3_c1_train_fwd": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 4 + [_P]),
    "danra_conv3x3_c1_train_dgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 4 + [_P]),
    "danra_conv3x3_c1_train_wgrad": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 4 + [_P, _P]),
    "danra_attention_small_supported": (ctypes.c_int, [ctypes.c_int] * 4),
    "danra_attention_small_fwd": (ctypes.c_int, [_P] * 4 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_small_bwd": (ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_small_qkv_fwd": (ctypes.c_int, [_P] * 2 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_small_qkv_bwd": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_core_qkv_fwd": (ctypes.c_int, [_P] * 3 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_core_qkv_bwd": (ctypes.c_int, [_P] * 6 + [ctypes.c_int] * 4 + [_P]),
    "danra_attention_core_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, _P]),
    "danra_batchnorm_train_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_int, _P, _P]),
    "danra_batchnorm_train_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, _P]),
    "danra_batchnorm_train_add_relu_fwd": (ctypes.c_int, [_P] * 9 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_float, _P, _P]),
    "danra_batchnorm_train_add_relu_bwd": (ctypes.c_int, [_P] * 10 + [ctypes.c_int] * 3 + [_P]),
    "danra_instancenorm_fwd": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, _P]),
    "danra_channel_sum_workspace": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "danra_channel_sum": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "danra_instancenorm_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    "danra_linear_wgrad_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "danra_linear_wgrad_workspace": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "danra_linear_wgrad": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "danra_linear_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "danra_linear": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_ln_tokens_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "danra_ln_tokens_workspace": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "danra_ln_tokens_fwd": (ctypes.c_int, [_P] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float, _P]),
    "danra_ln_tokens_bwd": (ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 3 + [_P, _P]),
    "danra_tokens_to_nchw_add": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 3 + [_P]),
    "danra_pointwise_train_supported": (ctypes.c_int, [ctypes.c_int] * 6),
    "danra_pointwise_train_workspace": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "danra_pointwise_train_fwd": (ctypes.c_int, [ctypes.c_int, _P, _P, _P, _P] + [ctypes.c_int] * 5 + [_P]),
    "danra_pointwise_train_dgrad": (ctypes.c_int, [ctypes.c_int, _P, _P, _P] + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_pointwise_train_wgrad": (ctypes.c_int, [ctypes.c_int, _P, _P, _P] + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3_train_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "danra_conv3x3_train_workspace": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "danra_conv3x3_train_fwd": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3_train_dgrad": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_conv3x3_train_wgrad": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 5 + [_P, _P]),
    "danra_set_graph_mode": (ctypes.c_int, [_P, ctypes.c_int]),
    "danra_set_first_image": (ctypes.c_int, [_P, ctypes.c_uint64]),
    "danra_profile_begin": (ctypes.c_int, [_P, ctypes.c_int]),
    "danra_profile_end": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong),
                                         ctypes.POINTER(ctypes.c_double)]),
    "danra_profile_executed": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "danra_profile_bytes": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double)]),
    "danra_tbias_width": (ctypes.c_int, [_P]),
    "danra_time_embed": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, _P, _P, _P, _P]),
    "danra_attention": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_conv1": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_conv2_bn_relu": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_resnet_layer": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_decoder_block": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_instance_norm": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_conv3x3": (ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 6 + [_P]),
    "danra_conv8x8s2": (ctypes.c_int, [_P] * 5 + [ctypes.c_int] * 6 + [_P]),
    "danra_conv3x3s2": (ctypes.c_int, [_P] * 5 + [ctypes.c_int] * 7 + [_P]),
    "danra_instance_norm_noalloc": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_edm_perturb": (ctypes.c_int, [_P] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float] * 3 + [ctypes.c_int, ctypes.c_uint64,
                                                                                              ctypes.c_uint64, _P]),
    "danra_edm_precondition": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_int, _P]),
    "danra_edm_dynamic_threshold_workspace": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_size_t]),
    "danra_edm_dynamic_threshold": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, _P, _P]),
    "danra_edm_euler": (ctypes.c_int, [_P] * 7 + [ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                                  ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
    "danra_edm_heun": (ctypes.c_int, [_P] * 7 + [ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                                 ctypes.c_float, ctypes.c_float, _P]),
    "danra_edm_renoise": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_float, ctypes.c_int, ctypes.c_uint64,
                                         ctypes.c_uint64, _P]),
    "danra_edm_finish": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 4 + [_P]),
    "danra_nchw_to_nhwc": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "danra_nhwc_to_nchw": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path():
    return _LIB_PATH


def load_library():
    """Load libdanra_hip.so once; raise RuntimeError if it was not built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(
                    f"{_LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                    "(or `python -c 'import __graft_entry__ as g; g.build()'`). The HIP path has no fallback.")
            lib = ctypes.CDLL(_LIB_PATH)
            for name, (restype, argtypes) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.danra_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name} must be a CUDA float32 tensor")
    return t if t.is_contiguous() else t.contiguous()


def _i64(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    t = t.to(torch.int64)
    return t if t.is_contiguous() else t.contiguous()


def _check_labels(y, num_classes):
    """The reference's nn.Embedding refuses an out-of-range label; the kernels clamp it (memory safety), so the
    refusal happens here (one device read per call)."""
    if y is not None and num_classes > 0 and y.numel():
        lo, hi = int(y.min()), int(y.max())
        if lo < 0 or hi >= num_classes:
            raise IndexError(f"label {lo if lo < 0 else hi} is out of range for the {num_classes}-class label embedding")


def reference_freq_tables(dim):
    """The two frequency tables evaluated with the torch CPU expressions of the reference
    (modules_DANRA_conditional.py:204-207 and :57), so that they round exactly as it does."""
    enc = 1.0 / (1000 ** (torch.arange(0, dim, 2).float() / dim))
    dec = torch.tensor([10000 ** (2 * i / dim) for i in range(dim // 2)], dtype=torch.float64).float()
    return enc.contiguous(), dec.contiguous()


def noise_image(x, t, alpha_hat, noise=None, seed=0, offset=0):
    """DiffusionUtils.noiseImage on the device: returns (noisy, noise).  `noise` given = the Gaussian draws to use
    (parity mode); otherwise they are drawn in the kernel from (seed, offset) and returned."""
    lib = load_library()
    x = _f32(x, "x")
    B = x.shape[0]
    per_image = x[0].numel()
    tt = _i64(t, "t")
    ah = _f32(alpha_hat, "alpha_hat")
    noisy = torch.empty_like(x)
    if noise is None:
        mode, z = NOISE_PHILOX, torch.empty_like(x)
    else:
        mode, z = NOISE_HOST, _f32(noise, "noise")
    with torch.cuda.device(x.device):
        _check(lib, lib.danra_noise_image(_ptr(x), _ptr(tt), _ptr(ah), int(ah.numel()), _ptr(noisy), _ptr(z), B,
                                          ctypes.c_size_t(per_image), mode, ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                          ctypes.c_uint64(int(offset) & 0xFFFFFFFFFFFFFFFF), _stream()), "danra_noise_image")
        for keep in (tt, ah):
            keep.record_stream(torch.cuda.current_stream())
    return noisy, z


# Under autocast the autograd Functions below receive their tensor inputs cast to float32 and run with autocast off;
# their backward passes run the same way.
_custom_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_custom_bwd = torch.amp.custom_bwd(device_type="cuda")


class _AttentionCoreHD16(torch.autograd.Function):
    """softmax(q k^T / 4) v per head with head_dim 16, forward and backward in libdanra_hip.so
    (danra_attention_core_fwd / _bwd): the L x L score matrices exist only in registers in either direction.
    qkv: [3, B, heads, L, 16] contiguous -> tokens [B, L, heads*16]."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, qkv):
        lib = load_library()
        qkv = qkv.contiguous()
        _, B, heads, L, hd = qkv.shape
        out = torch.empty((B, L, heads * hd), device=qkv.device, dtype=torch.float32)
        lse2 = torch.empty((B * heads, L), device=qkv.device, dtype=torch.float32)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_core_fwd(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(out), _ptr(lse2), B,
                                                     heads, L, hd, _stream()), "danra_attention_core_fwd")
        ctx.save_for_backward(qkv, out, lse2)
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = load_library()
        qkv, out, lse2 = ctx.saved_tensors
        _, B, heads, L, hd = qkv.shape
        dout = _f32(dout, "dout")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse2)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_core_bwd(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(out), _ptr(dout),
                                                     _ptr(lse2), _ptr(dqkv[0]), _ptr(dqkv[1]), _ptr(dqkv[2]), _ptr(delta),
                                                     B, heads, L, hd, _stream()), "danra_attention_core_bwd")
        return dqkv


class _AttentionCoreHD16Qkv(torch.autograd.Function):
    """The same core on the fused QKV projection's output used in place (danra_attention_core_qkv_fwd / _bwd): qkv
    [B, L, 3 * heads * 16] contiguous, as nn.MultiheadAttention's in_proj leaves it -> tokens [B, L, heads*16]; the gradient
    comes back in the same [B, L, 3 * heads * 16] layout, ready for the projection's backward - no copy into per-head tensors
    in either direction."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, qkv, heads):
        lib = load_library()
        qkv = qkv.contiguous()
        B, L, C3 = qkv.shape
        hd = C3 // (3 * heads)
        out = torch.empty((B, L, heads * hd), device=qkv.device, dtype=torch.float32)
        lse2 = torch.empty((B * heads, L), device=qkv.device, dtype=torch.float32)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_core_qkv_fwd(_ptr(qkv), _ptr(out), _ptr(lse2), B, heads, L, hd, _stream()),
                   "danra_attention_core_qkv_fwd")
        ctx.save_for_backward(qkv, out, lse2)
        ctx.heads = heads
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = load_library()
        qkv, out, lse2 = ctx.saved_tensors
        B, L, C3 = qkv.shape
        heads = ctx.heads
        dout = _f32(dout, "dout")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse2)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_core_qkv_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse2), _ptr(dqkv), _ptr(delta),
                                                         B, heads, L, C3 // (3 * heads), _stream()),
                   "danra_attention_core_qkv_bwd")
        return dqkv, None


class _AttentionCoreSmall(torch.autograd.Function):
    """softmax(q k^T / sqrt(head_dim)) v per head for the short sequences of the deep levels (at most 64 tokens, head_dim up
    to 128), forward and backward in libdanra_hip.so (danra_attention_small_fwd / _bwd): one workgroup per (image, head),
    the backward forms the probabilities again instead of keeping them.
    qkv: [3, B, heads, L, head_dim] contiguous -> tokens [B, L, heads*head_dim]."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, qkv):
        lib = load_library()
        qkv = qkv.contiguous()
        _, B, heads, L, hd = qkv.shape
        out = torch.empty((B, L, heads * hd), device=qkv.device, dtype=torch.float32)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_small_fwd(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(out), B, heads, L, hd,
                                                      _stream()), "danra_attention_small_fwd")
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = load_library()
        (qkv,) = ctx.saved_tensors
        _, B, heads, L, hd = qkv.shape
        dout = _f32(dout, "dout")
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_small_bwd(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(dout), _ptr(dqkv[0]),
                                                      _ptr(dqkv[1]), _ptr(dqkv[2]), B, heads, L, hd, _stream()),
                   "danra_attention_small_bwd")
        return dqkv


class _AttentionCoreSmallQkv(torch.autograd.Function):
    """The short-sequence core on the fused QKV projection's output used in place (danra_attention_small_qkv_fwd / _bwd):
    qkv [B, L, 3 * heads * head_dim] contiguous -> tokens [B, L, heads*head_dim]; the gradient comes back in qkv's layout."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, qkv, heads):
        lib = load_library()
        qkv = qkv.contiguous()
        B, L, C3 = qkv.shape
        hd = C3 // (3 * heads)
        out = torch.empty((B, L, heads * hd), device=qkv.device, dtype=torch.float32)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_small_qkv_fwd(_ptr(qkv), _ptr(out), B, heads, L, hd, _stream()),
                   "danra_attention_small_qkv_fwd")
        ctx.save_for_backward(qkv)
        ctx.heads = heads
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = load_library()
        (qkv,) = ctx.saved_tensors
        B, L, C3 = qkv.shape
        heads = ctx.heads
        dout = _f32(dout, "dout")
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _check(lib, lib.danra_attention_small_qkv_bwd(_ptr(qkv), _ptr(dout), _ptr(dqkv), B, heads, L, C3 // (3 * heads),
                                                          _stream()), "danra_attention_small_qkv_bwd")
        return dqkv, None


def attention_small_supported(tokens, n_heads):
    """True when the differentiable self-attention of [B, L, C] `tokens` can run its core in the native library's
    short-sequence kernels: a CUDA tensor (float32; half precision under autocast), at most 64 tokens."""
    B, L, C = tokens.shape
    return (native_training_kernels and tokens.is_cuda and _native_dtype(tokens) and C % n_heads == 0
            and bool(load_library().danra_attention_small_supported(B, n_heads, L, C // n_heads)))


def _native_dtype(x):
    """float32 always; under autocast (the reference's training loop runs its forward inside `autocast()`,
    training_DANRA_conditional.py:962-971) also the half-precision tensors autocast hands over: the native kernels
    then take them cast to float32 and return float32, which autocast's next operator casts down again as it would
    any float32 input."""
    return x.dtype == torch.float32 or (torch.is_autocast_enabled() and x.dtype in (torch.float16, torch.bfloat16))


def attention_core_supported(tokens, n_heads):
    """True when the differentiable self-attention of [B, L, C] `tokens` can run its core in the native library:
    a CUDA tensor (float32; half precision under autocast), head_dim 16, L a multiple of 128 (the 64-channel levels
    of the 4-head network)."""
    return (native_training_kernels and tokens.is_cuda and _native_dtype(tokens)
            and tokens.shape[-1] == 16 * n_heads and tokens.shape[1] % 128 == 0
            and tokens.shape[0] * n_heads * tokens.shape[1] < (1 << 27))


# Token counts from which the forward product and the input gradient of a projection go through danra_linear: the tall
# products of the 64-channel levels (8192 ... 131072 rows).  Below, the BLAS library's small kernels take 5-8 us where the
# implicit-GEMM kernel takes 11.
_NATIVE_LINEAR_MIN_ROWS = 4096


class _TokenLinear(torch.autograd.Function):
    """torch.nn.functional.linear(x, weight, bias) on [..., K] fp32 CUDA tokens, forward and backward in libdanra_hip.so:
    the product and the input gradient through danra_linear (the implicit-GEMM kernel with fp32 matrix instructions; for
    these tall, narrow products - 8192 ... 131072 rows, 64 ... 192 columns - the BLAS library behind torch.mm picks tiles of
    256 x 256 and takes 60-130 us per call on MI355X), weight and bias gradient through danra_linear_wgrad (a reduction
    over every token of the batch into a few thousand values)."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight, bias):
        lib = load_library()
        N, K = weight.shape
        x2 = x.reshape(-1, K).contiguous()
        weight = weight.contiguous()
        M = x2.shape[0]
        if M >= _NATIVE_LINEAR_MIN_ROWS:
            y = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
            with torch.cuda.device(x.device):
                _check(lib, lib.danra_linear(_ptr(x2), _ptr(weight), _ptr(bias), _ptr(y), M, N, K, _stream()), "danra_linear")
        else:
            y = torch.nn.functional.linear(x2, weight, bias).view(x.shape[:-1] + (N,))
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = _f32(dy, "dy").reshape(-1, N)
        M = x2.shape[0]
        dx = None
        with torch.cuda.device(x2.device):
            if ctx.needs_input_grad[0]:
                if M >= _NATIVE_LINEAR_MIN_ROWS and lib.danra_linear_supported(M, K, N):  # dx = dy W = dy (W^T)^T
                    dx = torch.empty(ctx.x_shape, device=x2.device, dtype=torch.float32)
                    wt = weight.t().contiguous()
                    _check(lib, lib.danra_linear(_ptr(dy2), _ptr(wt), None, _ptr(dx), M, K, N, _stream()), "danra_linear")
                else:
                    dx = dy2.matmul(weight).view(ctx.x_shape)
            if not ctx.needs_input_grad[1]:
                dw = None
                db = dy2.sum(0) if ctx.needs_input_grad[2] else None
            elif lib.danra_linear_wgrad_supported(M, N, K):
                dw = torch.empty_like(weight)
                db = torch.empty(N, device=x2.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
                workspace = torch.empty(lib.danra_linear_wgrad_workspace(M, N, K) // 4, device=x2.device, dtype=torch.float32)
                _check(lib, lib.danra_linear_wgrad(_ptr(dy2), _ptr(x2), _ptr(dw), _ptr(db) if db is not None else None, M, N, K,
                                                   _ptr(workspace), _stream()), "danra_linear_wgrad")
            else:
                # wider layers (the 128 ... 512-channel attention blocks, a few thousand tokens at most): dw = dy^T x is
                # danra_linear on the two transposed operands.  (The BLAS library behind torch.mm runs the 256 x 256 result
                # of the 256-channel blocks as ONE 256 x 256 tile on one workgroup: 121 us per call on MI355X.)
                # (a long reduction over few output tiles stays torch's: with the limit at 2048 rows the whole training step
                # measured 5.18 against 5.00 ms)
                if M <= 512 and lib.danra_linear_supported(N, K, M):
                    dw = torch.empty_like(weight)
                    dyt, xt = dy2.t().contiguous(), x2.t().contiguous()
                    _check(lib, lib.danra_linear(_ptr(dyt), _ptr(xt), None, _ptr(dw), N, K, M, _stream()), "danra_linear")
                else:
                    dw = dy2.t().mm(x2)
                db = dy2.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def token_linear(x, weight, bias):
    """``torch.nn.functional.linear`` through the native library (see _TokenLinear): on CUDA, float32 parameters, N and K
    multiples of 64 (every projection of the network's attention blocks), with the native training kernels enabled;
    torch's own operator otherwise."""
    if (native_training_kernels and x.is_cuda and weight.dtype == torch.float32 and weight.requires_grad
            and bias is not None and torch.is_grad_enabled()):
        N, K = weight.shape
        M = x.numel() // K
        if load_library().danra_linear_supported(M, N, K) and load_library().danra_linear_supported(M, K, N):
            return _TokenLinear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def differentiable_self_attention(mha, tokens):
    """nn.MultiheadAttention(tokens, tokens, tokens, need_weights=False)[0] for a batch_first, dropout-free module: the
    two projections through token_linear, the scaled-dot-product core in the native library where it has the shape
    (attention_core_supported: head_dim 16, the 64-channel levels - there the core reads the projection's output where it
    lies; attention_small_supported: the short sequences of the deep levels) and torch's fused scaled_dot_product_attention otherwise."""
    B, L, C = tokens.shape
    heads = mha.num_heads
    qkv = token_linear(tokens, mha.in_proj_weight, mha.in_proj_bias)
    if attention_core_supported(tokens, heads):        # (both native cores read the projection's [B, L, 3 C] output in place)
        attended = _AttentionCoreHD16Qkv.apply(qkv, heads)
    elif attention_small_supported(tokens, heads):
        attended = _AttentionCoreSmallQkv.apply(qkv, heads)
    else:
        qkv = qkv.view(B, L, 3, heads, C // heads).permute(2, 0, 3, 1, 4)  # [3, B, heads, L, head_dim]
        attended = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        attended = attended.permute(0, 2, 1, 3).reshape(B, L, C)
    return token_linear(attended, mha.out_proj.weight, mha.out_proj.bias)


def differentiable_attention_supported(mha, tokens):
    """True when differentiable_self_attention can take the place of `mha(tokens, tokens, tokens)`: a CUDA tensor
    (float32; half precision under autocast) and projections the library's linear kernel has."""
    if not (native_training_kernels and tokens.is_cuda and _native_dtype(tokens) and mha.in_proj_bias is not None
            and mha.in_proj_weight is not None and mha.in_proj_weight.dtype == torch.float32):
        return False
    B, L, C = tokens.shape
    lib = load_library()
    return bool(lib.danra_linear_supported(B * L, 3 * C, C) and lib.danra_linear_supported(B * L, C, 3 * C)
                and lib.danra_linear_supported(B * L, C, C))


class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in train mode (+ optional ReLU) on a contiguous NCHW fp32 CUDA tensor, forward and backward in
    libdanra_hip.so (danra_batchnorm_train_fwd / _bwd).  The running statistics and the int64 counter `batches`
    (num_batches_tracked) are updated in place by the kernel."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, batches):
        lib = load_library()
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(C, device=x.device, dtype=torch.float32)
        save_rstd = torch.empty(C, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_batchnorm_train_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(save_mean),
                                                      _ptr(save_rstd), _ptr(running_mean), _ptr(running_var), B, C, H * W,
                                                      ctypes.c_float(eps), ctypes.c_float(momentum), int(relu), _ptr(batches),
                                                      _stream()), "danra_batchnorm_train_fwd")
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, y if relu else None, gamma, save_mean, save_rstd)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, y, gamma, save_mean, save_rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _f32(dy, "dy")
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_batchnorm_train_bwd(_ptr(x), _ptr(dy), _ptr(y), _ptr(gamma), _ptr(save_mean),
                                                      _ptr(save_rstd), _ptr(dx), _ptr(dgamma), _ptr(dbeta), B, C, H * W,
                                                      int(ctx.relu), _stream()), "danra_batchnorm_train_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None


class _BatchNormAddReluTrain(torch.autograd.Function):
    """relu(bn(x) + res) for a train-mode nn.BatchNorm2d - the tail of a BasicBlock - as one kernel in either direction
    (danra_batchnorm_train_add_relu_fwd / _bwd) on contiguous NCHW fp32 CUDA tensors."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, eps, momentum, batches):
        lib = load_library()
        x, res = x.contiguous(), res.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(C, device=x.device, dtype=torch.float32)
        save_rstd = torch.empty(C, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_batchnorm_train_add_relu_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(y),
                                                               _ptr(save_mean), _ptr(save_rstd), _ptr(running_mean),
                                                               _ptr(running_var), B, C, H * W, ctypes.c_float(eps),
                                                               ctypes.c_float(momentum), _ptr(batches), _stream()),
                   "danra_batchnorm_train_add_relu_fwd")
        ctx.save_for_backward(x, y, gamma, save_mean, save_rstd)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, y, gamma, save_mean, save_rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _f32(dy, "dy")
        dx, dres = torch.empty_like(x), torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_batchnorm_train_add_relu_bwd(_ptr(x), _ptr(dy), _ptr(y), _ptr(gamma), _ptr(save_mean),
                                                               _ptr(save_rstd), _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta),
                                                               B, C, H * W, _stream()), "danra_batchnorm_train_add_relu_bwd")
        return dx, dres, dgamma, dbeta, None, None, None, None, None


def _batch_counter(bn):
    """bn.num_batches_tracked where the kernel can increment it (an int64 scalar on the device), else it is incremented
    here and None is returned."""
    n = bn.num_batches_tracked
    if n is not None and n.is_cuda and n.dtype == torch.int64 and n.numel() == 1 and n.is_contiguous():
        return n
    if n is not None:
        with torch.no_grad():
            n.add_(1)
    return None


def batchnorm_add_relu_train(bn, x, res):
    """relu(bn(x) + res) for a train-mode nn.BatchNorm2d through the native kernel (see batchnorm_train_supported; `res` has
    x's shape)."""
    return _BatchNormAddReluTrain.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                                        float(bn.momentum), _batch_counter(bn))


def batchnorm_train_supported(bn, x):
    """True when `bn(x)` in train mode can run in the native library: affine BatchNorm2d with running statistics and a
    fixed momentum, on a 4-D CUDA tensor (float32; half precision under autocast) whose H*W is a multiple of 4.  A
    tensor that is not contiguous NCHW (a convolution fed by an attention block's output hands over channels-last
    strides) is copied first."""
    return (native_training_kernels and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and x.is_cuda
            and _native_dtype(x) and x.dim() == 4 and bn.weight.dtype == torch.float32
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] * x.shape[2] * x.shape[3] >= 2 and x.numel() < (1 << 31))


def batchnorm_train(bn, x, relu=False):
    """bn(x) [then ReLU] for a train-mode nn.BatchNorm2d through the native kernels (see batchnorm_train_supported)."""
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                                 float(bn.momentum), bool(relu), _batch_counter(bn))


class _InstanceNorm(torch.autograd.Function):
    """nn.InstanceNorm2d(affine=False) on a contiguous NCHW fp32 CUDA tensor, forward and backward in libdanra_hip.so
    (danra_instancenorm_fwd / _bwd)."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, eps):
        lib = load_library()
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(B * C, device=x.device, dtype=torch.float32)
        save_rstd = torch.empty(B * C, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_instancenorm_fwd(_ptr(x), _ptr(y), _ptr(save_mean), _ptr(save_rstd), B * C, H * W,
                                                   ctypes.c_float(eps), _stream()), "danra_instancenorm_fwd")
        ctx.save_for_backward(x, save_mean, save_rstd)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, save_mean, save_rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _f32(dy, "dy")
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_instancenorm_bwd(_ptr(x), _ptr(dy), _ptr(save_mean), _ptr(save_rstd), _ptr(dx), B * C,
                                                   H * W, _stream()), "danra_instancenorm_bwd")
        return dx, None


def instance_norm(module, x):
    """`module(x)`; an nn.InstanceNorm2d without affine parameters or running statistics on a 4-D CUDA tensor (float32;
    half precision under autocast; H*W a multiple of 4) runs forward and backward in the native library."""
    if (native_training_kernels and type(module) is torch.nn.InstanceNorm2d and not module.affine and not module.track_running_stats and x.is_cuda
            and _native_dtype(x) and x.dim() == 4
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.numel() < (1 << 31)):
        return _InstanceNorm.apply(x, float(module.eps))
    return module(x)


class _Conv3x3Train(torch.autograd.Function):
    """torch.nn.functional.conv2d(x, weight, None, stride=1, padding=1) for a 3x3 kernel on a contiguous NCHW fp32 CUDA
    tensor, forward, data gradient and weight gradient in libdanra_hip.so (danra_conv3x3_train_fwd / _dgrad / _wgrad):
    the tensors and the [Cout][Cin][3][3] weight are used where they lie, no layout conversion around a call."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight):
        lib = load_library()
        x = x.contiguous()
        weight = weight.contiguous()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
        ws = torch.empty(max(lib.danra_conv3x3_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_conv3x3_train_fwd(_ptr(x), _ptr(weight), _ptr(y), B, Cin, Cout, H, W, _ptr(ws), _stream()),
                   "danra_conv3x3_train_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = _f32(dy, "dy")
        ws = torch.empty(max(lib.danra_conv3x3_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=x.device, dtype=torch.float32)
        dx = dw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(lib, lib.danra_conv3x3_train_dgrad(_ptr(dy), _ptr(weight), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws),
                                                          _stream()), "danra_conv3x3_train_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                _check(lib, lib.danra_conv3x3_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws),
                                                          _stream()), "danra_conv3x3_train_wgrad")
        return dx, dw


class _Conv3x3C1Train(torch.autograd.Function):
    """torch.nn.functional.conv2d(x, weight, None, stride=1, padding=1) for a [1][Cin][3][3] weight (the network's last
    convolution) on a contiguous NCHW fp32 CUDA tensor: forward, data gradient and weight gradient in libdanra_hip.so
    (danra_conv3x3_c1_train_fwd / _dgrad / _wgrad)."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight):
        lib = load_library()
        x = x.contiguous()
        weight = weight.contiguous()
        B, Cin, H, W = x.shape
        y = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_conv3x3_c1_train_fwd(_ptr(x), _ptr(weight), _ptr(y), B, Cin, H, W, _stream()),
                   "danra_conv3x3_c1_train_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        dy = _f32(dy, "dy")
        dx = dw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(lib, lib.danra_conv3x3_c1_train_dgrad(_ptr(dy), _ptr(weight), _ptr(dx), B, Cin, H, W, _stream()),
                       "danra_conv3x3_c1_train_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                ws = torch.empty(max(lib.danra_conv3x3_c1_train_workspace(B, Cin, H, W) // 4, 1), device=x.device, dtype=torch.float32)
                _check(lib, lib.danra_conv3x3_c1_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, H, W, _ptr(ws), _stream()),
                       "danra_conv3x3_c1_train_wgrad")
        return dx, dw


class _Conv3x3S2Train(torch.autograd.Function):
    """torch.nn.functional.conv2d(x, weight, None, stride=2, padding=1) for a 3x3 kernel on a contiguous NCHW fp32 CUDA tensor
    with even sides: forward, data gradient and weight gradient in libdanra_hip.so (danra_conv3x3s2_train_fwd / _dgrad /
    _wgrad), tensors and weight used where they lie."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight):
        lib = load_library()
        x = x.contiguous()
        weight = weight.contiguous()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B, Cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
        ws = torch.empty(max(lib.danra_conv3x3s2_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_conv3x3s2_train_fwd(_ptr(x), _ptr(weight), _ptr(y), B, Cin, Cout, H, W, _ptr(ws), _stream()),
                   "danra_conv3x3s2_train_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = _f32(dy, "dy")
        ws = torch.empty(max(lib.danra_conv3x3s2_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=x.device, dtype=torch.float32)
        dx = dw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(lib, lib.danra_conv3x3s2_train_dgrad(_ptr(dy), _ptr(weight), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws),
                                                            _stream()), "danra_conv3x3s2_train_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                _check(lib, lib.danra_conv3x3s2_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws),
                                                            _stream()), "danra_conv3x3s2_train_wgrad")
        return dx, dw


class _Conv8x8S2Train(torch.autograd.Function):
    """torch.nn.functional.conv2d(x, weight, None, stride=2, padding=3) for an 8x8 kernel (the two stem convolutions of the
    encoder) on a contiguous NCHW fp32 CUDA tensor: forward, data gradient and weight gradient in libdanra_hip.so
    (danra_conv8x8s2_train_fwd / _dgrad / _wgrad), tensors and weight used where they lie."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight):
        lib = load_library()
        x = x.contiguous()
        weight = weight.contiguous()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B, Cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_conv8x8s2_train_fwd(_ptr(x), _ptr(weight), _ptr(y), B, Cin, Cout, H, W, _stream()),
                   "danra_conv8x8s2_train_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = _f32(dy, "dy")
        dx = dw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(lib, lib.danra_conv8x8s2_train_dgrad(_ptr(dy), _ptr(weight), _ptr(dx), B, Cin, Cout, H, W, _stream()),
                       "danra_conv8x8s2_train_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                ws = torch.empty(max(lib.danra_conv8x8s2_train_workspace(B, Cin, Cout, H, W) // 4, 1), device=x.device,
                                 dtype=torch.float32)
                _check(lib, lib.danra_conv8x8s2_train_wgrad(_ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws), _stream()),
                       "danra_conv8x8s2_train_wgrad")
        return dx, dw


class _LNTokens(torch.autograd.Function):
    """x [B, C, H, W] -> LayerNorm_C(x.view(B, C, H*W).transpose(1, 2)) as [B, H*W, C]: the entry of an attention block,
    forward and backward as one kernel each in libdanra_hip.so (danra_ln_tokens_fwd / _bwd): no transposed copy of the map."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, gamma, beta, eps):
        lib = load_library()
        x = x.contiguous()
        B, C, H, W = x.shape
        L = H * W
        normed = torch.empty((B, L, C), device=x.device, dtype=torch.float32)
        mean = torch.empty(B * L, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_ln_tokens_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(normed), _ptr(mean), _ptr(rstd), B, C, L,
                                                ctypes.c_float(eps), _stream()), "danra_ln_tokens_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return normed

    @staticmethod
    @_custom_bwd
    def backward(ctx, dn):
        lib = load_library()
        x, gamma, mean, rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        L = H * W
        dn = _f32(dn, "dn")
        dx = torch.empty_like(x)
        dgb = torch.empty((2, C), device=x.device, dtype=torch.float32)
        ws = torch.empty(max(lib.danra_ln_tokens_workspace(B, C, L) // 4, 1), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_ln_tokens_bwd(_ptr(dn), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dgb), B, C,
                                                L, _ptr(ws), _stream()), "danra_ln_tokens_bwd")
        return dx, dgb[0], dgb[1], None


class _TokensResidualNCHW(torch.autograd.Function):
    """(tokens [B, L, C], x [B, C, H, W]) -> x + tokens.transpose(1, 2).view_as(x): the exit of an attention block as one
    kernel (danra_tokens_to_nchw_add); the backward pass is one transposition (danra_nchw_to_nhwc)."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, tokens, x):
        lib = load_library()
        tokens, x = tokens.contiguous(), x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_tokens_to_nchw_add(_ptr(tokens), _ptr(x), _ptr(out), B, C, H * W, _stream()),
                   "danra_tokens_to_nchw_add")
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        lib = load_library()
        dout = _f32(dout, "dout")
        B, C, H, W = dout.shape
        dtok = torch.empty((B, H * W, C), device=dout.device, dtype=torch.float32)
        with torch.cuda.device(dout.device):
            _check(lib, lib.danra_nchw_to_nhwc(_ptr(dout), _ptr(dtok), B, C, H * W, _stream()), "danra_nchw_to_nhwc")
        return dtok, dout


def attention_block_supported(layernorm, x):
    """True when the entry and exit of an attention block on the NCHW map `x` can run in the native library: the training
    step on a CUDA tensor (float32; half precision under autocast), an affine nn.LayerNorm over the channels, a token and
    channel count the kernels have (danra_ln_tokens_supported), a batch the transposition's grid takes."""
    if not (native_training_kernels and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4 and _native_dtype(x)
            and type(layernorm) is torch.nn.LayerNorm and layernorm.elementwise_affine and layernorm.bias is not None
            and layernorm.weight.dtype == torch.float32 and tuple(layernorm.normalized_shape) == (x.shape[1],)
            and x.shape[0] <= 65535):
        return False
    return bool(load_library().danra_ln_tokens_supported(x.shape[0], x.shape[1], x.shape[2] * x.shape[3]))


def ln_tokens(layernorm, x):
    """layernorm(x.reshape(B, C, H*W).permute(0, 2, 1)) through the native kernel (see attention_block_supported)."""
    return _LNTokens.apply(x, layernorm.weight, layernorm.bias, float(layernorm.eps))


def tokens_residual_nchw(tokens, x):
    """(tokens + x.reshape(B, C, H*W).permute(0, 2, 1)).permute(0, 2, 1).reshape_as(x) through the native kernel."""
    return _TokensResidualNCHW.apply(tokens, x)


def channel_sum(t):
    """t.sum((0, 2, 3)) of a 4-D fp32 CUDA tensor through the native kernel (contiguous NCHW, H*W a multiple of 4), torch's
    reduction otherwise."""
    if t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.is_contiguous() and (t.shape[2] * t.shape[3]) % 4 == 0 \
            and 0 < t.numel() < (1 << 31):
        lib = load_library()
        out = torch.empty(t.shape[1], device=t.device, dtype=torch.float32)
        B, C, HW = t.shape[0], t.shape[1], t.shape[2] * t.shape[3]
        need = lib.danra_channel_sum_workspace(B, C, HW)
        ws = torch.empty(need // 4, device=t.device, dtype=torch.float32) if need else None
        with torch.cuda.device(t.device):
            _check(lib, lib.danra_channel_sum(_ptr(t), _ptr(out), B, C, HW, _ptr(ws), _stream()), "danra_channel_sum")
        return out
    return t.sum((0, 2, 3))


class _AddChannelBias(torch.autograd.Function):
    """y + bias.view(1, -1, 1, 1) for a convolution's output: torch's add forward; backward the gradient itself for y and
    its per-channel sum (danra_channel_sum) for the bias."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, y, bias):
        return y + bias.view(1, -1, 1, 1)

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        return (dy if ctx.needs_input_grad[0] else None,
                channel_sum(dy) if ctx.needs_input_grad[1] else None)


def _with_bias(y, bias):
    return y if bias is None else _AddChannelBias.apply(y, bias)


PW_CONV1X1_S2, PW_CONVT2X2 = 0, 1


class _PointwiseTrain(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, 1, stride 2, bias=False) (kind 0) or nn.ConvTranspose2d(Cin, Cout, 2, stride 2) with its bias
    (kind 1) on a contiguous NCHW fp32 CUDA tensor: forward, data gradient and weight gradient in libdanra_hip.so
    (danra_pointwise_train_fwd / _dgrad / _wgrad); the bias gradient is torch's reduction."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, x, weight, bias, kind):
        lib = load_library()
        x = x.contiguous()
        weight = weight.contiguous()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0] if kind == PW_CONV1X1_S2 else weight.shape[1]
        shape = (B, Cout, H // 2, W // 2) if kind == PW_CONV1X1_S2 else (B, Cout, 2 * H, 2 * W)
        y = torch.empty(shape, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(lib, lib.danra_pointwise_train_fwd(kind, _ptr(x), _ptr(weight), _ptr(bias), _ptr(y), B, Cin, Cout, H, W,
                                                      _stream()), "danra_pointwise_train_fwd")
        ctx.save_for_backward(x, weight)
        ctx.kind, ctx.Cout, ctx.has_bias = kind, Cout, bias is not None
        return y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dy):
        lib = load_library()
        x, weight = ctx.saved_tensors
        kind, Cout = ctx.kind, ctx.Cout
        B, Cin, H, W = x.shape
        dy = _f32(dy, "dy")
        dx = dw = db = None
        with torch.cuda.device(x.device):
            ws = torch.empty(max(lib.danra_pointwise_train_workspace(kind, B, Cin, Cout, H, W) // 4, 1), device=x.device,
                             dtype=torch.float32)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _check(lib, lib.danra_pointwise_train_dgrad(kind, _ptr(dy), _ptr(weight), _ptr(dx), B, Cin, Cout, H, W, _ptr(ws),
                                                            _stream()), "danra_pointwise_train_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                _check(lib, lib.danra_pointwise_train_wgrad(kind, _ptr(x), _ptr(dy), _ptr(dw), B, Cin, Cout, H, W, _ptr(ws),
                                                            _stream()), "danra_pointwise_train_wgrad")
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = channel_sum(dy)
        return dx, dw, db, None


def _pointwise_ok(x, weight):
    return (native_training_kernels and native_training_convolutions and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4
            and _native_dtype(x) and weight.dtype == torch.float32 and (x.requires_grad or weight.requires_grad))


def conv_transpose2d(convt, x):
    """``convt(x)`` for an nn.ConvTranspose2d.  A kernel-2, stride-2, unpadded, ungrouped one on a 4-D CUDA tensor with
    gradients enabled - a decoder block's up-sampling in the training step - runs forward and backward in the native library
    where it has the shape (danra_pointwise_train_supported); everything else is the module itself."""
    if (type(convt) is torch.nn.ConvTranspose2d and convt.kernel_size == (2, 2) and convt.stride == (2, 2)
            and convt.padding == (0, 0) and convt.output_padding == (0, 0) and convt.dilation == (1, 1) and convt.groups == 1
            and _pointwise_ok(x, convt.weight)):
        B, Cin, H, W = x.shape
        if load_library().danra_pointwise_train_supported(PW_CONVT2X2, B, Cin, convt.out_channels, H, W):
            return _PointwiseTrain.apply(x, convt.weight, convt.bias, PW_CONVT2X2)
    return convt(x)


def conv2d(conv, x):
    """``conv(x)`` for an nn.Conv2d.  A 3x3, stride-1, padding-1, ungrouped convolution on a 4-D CUDA tensor (float32; half
    precision under autocast) with gradients enabled - the training step - runs forward and backward in the native library
    where it has the shape (danra_conv3x3_train_supported); the bias, if any, is added by torch.  So does the stride-2 form
    of that convolution and the 1x1 stride-2 shortcut beside it.  Everything else is the module itself."""
    if (native_training_kernels and native_training_convolutions and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4
            and type(conv) is torch.nn.Conv2d and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and _native_dtype(x)
            and conv.weight.dtype == torch.float32 and (x.requires_grad or conv.weight.requires_grad)):
        B, Cin, H, W = x.shape
        if load_library().danra_conv3x3_train_supported(B, Cin, conv.out_channels, H, W):
            y = _Conv3x3Train.apply(x, conv.weight)
            return _with_bias(y, conv.bias)
        if conv.out_channels == 1 and load_library().danra_conv3x3_c1_train_supported(B, Cin, H, W):   # the network's last convolution
            y = _Conv3x3C1Train.apply(x, conv.weight)
            return _with_bias(y, conv.bias)
    # 3x3, stride 2, padding 1 on even sides (the first convolution of a down-sampling BasicBlock): kernels of its own
    # (danra_conv3x3s2_train_*); a map they do not take (sides that are not powers of two) goes through the stride-1 kernels
    # evaluated everywhere with every second row and column kept, and autograd scatters dy back into the full map
    if (native_training_kernels and native_training_convolutions and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4
            and type(conv) is torch.nn.Conv2d and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and _native_dtype(x)
            and conv.weight.dtype == torch.float32 and (x.requires_grad or conv.weight.requires_grad)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
        B, Cin, H, W = x.shape
        if load_library().danra_conv3x3s2_train_supported(B, Cin, conv.out_channels, H, W):
            y = _Conv3x3S2Train.apply(x, conv.weight)
            return _with_bias(y, conv.bias)
        if load_library().danra_conv3x3_train_supported(B, Cin, conv.out_channels, H, W):
            y = _Conv3x3Train.apply(x, conv.weight)[:, :, ::2, ::2]
            return _with_bias(y, conv.bias)
    # 8x8, stride 2, padding 3, no bias: the two stem convolutions (an input that asks for a gradient needs the data-gradient
    # kernel, which exists for 32-channel multiples only)
    if (native_training_kernels and native_training_convolutions and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4
            and type(conv) is torch.nn.Conv2d and conv.kernel_size == (8, 8) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and conv.bias is None
            and _native_dtype(x) and conv.weight.dtype == torch.float32 and (x.requires_grad or conv.weight.requires_grad)):
        B, Cin, H, W = x.shape
        lib = load_library()
        if (lib.danra_conv8x8s2_train_dgrad_supported if x.requires_grad else lib.danra_conv8x8s2_train_supported)(
                B, Cin, conv.out_channels, H, W):
            return _Conv8x8S2Train.apply(x, conv.weight)
    if (type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (2, 2) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and _pointwise_ok(x, conv.weight)):
        B, Cin, H, W = x.shape
        if load_library().danra_pointwise_train_supported(PW_CONV1X1_S2, B, Cin, conv.out_channels, H, W):
            return _PointwiseTrain.apply(x, conv.weight, None, PW_CONV1X1_S2)
    return conv(x)


class UNetEngine:
    """One native handle = one model geometry on one device.  Borrows parameter pointers while
    packing, owns its packed-weight arena and activation workspace."""

    def __init__(self, *, has_lsm, has_topo, has_cond, num_classes, time_embedding, out_channels, n_heads,
                 block_layers, device, attn_ff=False):
        self.lib = load_library()
        self.device = torch.device(device)
        cfg = DanraConfig(int(has_lsm), int(has_topo), int(has_cond), int(num_classes or 0), int(time_embedding),
                          int(out_channels), int(n_heads), (ctypes.c_int * 4)(*block_layers), int(bool(attn_ff)))
        self.cfg = cfg
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.danra_create(ctypes.byref(cfg), ctypes.byref(self.handle)), "danra_create")
        enc, dec = reference_freq_tables(int(time_embedding))
        _check(self.lib, self.lib.danra_set_freq_tables(self.handle, _ptr(enc), _ptr(dec), enc.numel()),
               "danra_set_freq_tables")
        self.out_channels = int(out_channels)
        self.packed_signature = None

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.danra_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def param_names(self):
        n = self.lib.danra_num_params(self.handle)
        return [(self.lib.danra_param_name(self.handle, i).decode(), self.lib.danra_param_numel(self.handle, i))
                for i in range(n)]

    def pack(self, state_dict):
        """Register every arithmetic-carrying state_dict entry and repack the arena."""
        keep = []
        with torch.cuda.device(self.device):
            for name, numel in self.param_names():
                if name not in state_dict:
                    raise RuntimeError(f"state_dict entry '{name}' is missing")
                t = state_dict[name].detach()
                if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.to(device=self.device, dtype=torch.float32).contiguous()
                keep.append(t)
                _check(self.lib, self.lib.danra_set_param(self.handle, name.encode(), _ptr(t), t.numel()),
                       f"danra_set_param({name})")
            _check(self.lib, self.lib.danra_pack_weights(self.handle, _stream()), "danra_pack_weights")
            torch.cuda.current_stream().synchronize()  # `keep` may be freed after this
        del keep

    def set_graph_mode(self, enable):
        """True (default): sample() replays one captured hipGraph per denoise step; False: every kernel of
        every step is enqueued.  Same kernels, bit-identical results."""
        _check(self.lib, self.lib.danra_set_graph_mode(self.handle, int(bool(enable))), "danra_set_graph_mode")

    # ------------------------------------------------------------------ measurement
    def profile_begin(self, kind=1):
        _check(self.lib, self.lib.danra_profile_begin(self.handle, kind), "danra_profile_begin")

    def profile_end(self):
        """(total kernel milliseconds, launches, algorithmic FLOPs) of the launches since profile_begin."""
        ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        _check(self.lib, self.lib.danra_profile_end(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)),
               "danra_profile_end")
        return ms.value, n.value, fl.value

    def profile_executed(self):
        """(FLOPs executed by the issued matrix instructions, algorithmic FLOPs that ran in the split-operand fp16 form)
        of the launches counted by the last profile_begin / profile_end pair."""
        ex, sp = ctypes.c_double(), ctypes.c_double()
        _check(self.lib, self.lib.danra_profile_executed(self.handle, ctypes.byref(ex), ctypes.byref(sp)),
               "danra_profile_executed")
        return ex.value, sp.value

    def profile_bytes(self):
        """Algorithmic bytes (every operand, weight and result once, 4 bytes an element) of the launches counted by the last
        profile_begin / profile_end pair."""
        b = ctypes.c_double()
        _check(self.lib, self.lib.danra_profile_bytes(self.handle, ctypes.byref(b)), "danra_profile_bytes")
        return b.value

    # ------------------------------------------------------------------ hot path
    def forward(self, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None):
        x = _f32(x, "x")
        B, _, H, W = x.shape
        _check_labels(y, self.cfg.num_classes)
        out = torch.empty((B, self.out_channels, H, W), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _check(self.lib, self.lib.danra_unet_forward(
                self.handle, _ptr(x), _ptr(_i64(t, "t")), _ptr(_i64(y, "y")), _ptr(_f32(cond_img, "cond_img")),
                _ptr(_f32(lsm_cond, "lsm_cond")), _ptr(_f32(topo_cond, "topo_cond")), _ptr(out), B, H, W, _stream()),
                "danra_unet_forward")
        self._last_shape = (B, H, W, x.device)
        return out

    def fmaps(self):
        """The five encoder feature maps of the last forward (Encoder.forward's return value), NCHW."""
        B, H, W, dev = self._last_shape
        outs = []
        side_h, side_w = H // 2, W // 2
        for ch in (64, 64, 128, 256, 512):
            outs.append(torch.empty((B, ch, side_h, side_w), device=dev, dtype=torch.float32))
            side_h //= 2
            side_w //= 2
        with torch.cuda.device(dev):
            _check(self.lib, self.lib.danra_get_fmaps(self.handle, *[_ptr(o) for o in outs], _stream()),
                   "danra_get_fmaps")
        return tuple(outs)

    def sample(self, x, y, cond_img, lsm_cond, topo_cond, betas, alphas, alpha_hat, noise=None, seed=0,
               trace_steps=None, cfg_scale=0.0, noise_scale=1.0, first_image=0):
        """Whole reverse loop on the device.  x is updated in place and returned.  `noise` (a CUDA
        tensor [(T-2), B, 1, H, W]) selects host-noise mode, otherwise Philox noise from `seed`.
        cfg_scale > 0: classifier-free guidance (two evaluations per step); noise_scale multiplies every draw.
        first_image: index of x[0] in the global batch this call samples a slice of (Philox draws are per global element)."""
        x = _f32(x, "x")
        B, _, H, W = x.shape
        _check_labels(y, self.cfg.num_classes)
        T = int(betas.numel())
        hb = betas.detach().to("cpu", torch.float32).contiguous()
        ha = alphas.detach().to("cpu", torch.float32).contiguous()
        hh = alpha_hat.detach().to("cpu", torch.float32).contiguous()
        mode = NOISE_PHILOX if noise is None else NOISE_HOST
        noise = _f32(noise, "noise")
        trace = None
        steps_arr = None
        n_trace = 0
        if trace_steps:
            n_trace = len(trace_steps)
            steps_arr = (ctypes.c_int * n_trace)(*trace_steps)
            trace = torch.empty((n_trace,) + tuple(x.shape), device=x.device, dtype=torch.float32)
        yy = _i64(y, "y")
        cc, ll, tt = _f32(cond_img, "cond_img"), _f32(lsm_cond, "lsm_cond"), _f32(topo_cond, "topo_cond")
        _check(self.lib, self.lib.danra_set_first_image(self.handle, ctypes.c_uint64(int(first_image))),
               "danra_set_first_image")
        with torch.cuda.device(x.device):
            _check(self.lib, self.lib.danra_sample_guided(
                self.handle, _ptr(x), _ptr(yy), _ptr(cc), _ptr(ll), _ptr(tt), B, H, W, T, _ptr(hb), _ptr(ha), _ptr(hh),
                mode, _ptr(noise), ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), ctypes.c_float(float(cfg_scale)),
                ctypes.c_float(float(noise_scale)),
                ctypes.cast(steps_arr, ctypes.c_void_p) if steps_arr is not None else None, n_trace, _ptr(trace),
                _stream()), "danra_sample_guided")
            # the launches above borrow yy/cc/ll/tt/noise: keep them alive until the stream has run them
            for keep in (yy, cc, ll, tt, noise):
                if keep is not None:
                    keep.record_stream(torch.cuda.current_stream())
        return (x, trace) if trace_steps else x


# ---------------------------------------------------------------------------------------------------------------------
# The steps of the elucidated (EDM) sampler (CascadedDDPM/elucidated_imagen.py:393-545), thin wrappers of the C entries.
# Images are contiguous CUDA float32 NCHW tensors; `mask` is a contiguous CUDA uint8 tensor [B, H, W] (or [B, 1, H, W]).
def _u64(v):
    return ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)


def _edm_plain(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name} must be a contiguous CUDA float32 tensor")
    return t


def _edm_mask(mask, images):
    if mask is None:
        return None
    B, HW = images.shape[0], images.shape[-1] * images.shape[-2]
    if not (mask.is_cuda and mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == B * HW):
        raise RuntimeError("mask must be a contiguous CUDA uint8 tensor with one value per image and pixel")
    return mask


def _edm_dims(images):
    if images.dim() != 4:
        raise RuntimeError("images must be [B, C, H, W]")
    B, C, H, W = images.shape
    return B, C, H * W


def edm_perturb(images, images_hat, net_in, s_noise, noise_scale, c_in, noise=None, seed=0, offset=0, inpaint=None,
                mask=None):
    lib = load_library()
    B, C, HW = _edm_dims(_edm_plain(images, "images"))
    with torch.cuda.device(images.device):
        _check(lib, lib.danra_edm_perturb(
            _ptr(images), _ptr(_edm_plain(noise, "noise")), _ptr(_edm_plain(inpaint, "inpaint")),
            _ptr(_edm_mask(mask, images)), _ptr(_edm_plain(images_hat, "images_hat")), _ptr(_edm_plain(net_in, "net_in")),
            B, C, HW, s_noise, noise_scale, c_in, NOISE_PHILOX if noise is None else NOISE_HOST, _u64(seed), _u64(offset),
            _stream()), "danra_edm_perturb")


def edm_precondition(x, net_out, model_out, c_skip, c_out, clamp):
    lib = load_library()
    with torch.cuda.device(x.device):
        _check(lib, lib.danra_edm_precondition(_ptr(_edm_plain(x, "x")), _ptr(_edm_plain(net_out, "net_out")),
                                               _ptr(_edm_plain(model_out, "model_out")), ctypes.c_size_t(x.numel()), c_skip,
                                               c_out, int(bool(clamp)), _stream()), "danra_edm_precondition")


def edm_dynamic_threshold(x, percentile, out=None):
    """s[b] = max(quantile(|x[b]|, percentile), 1): the reference's threshold_x_start up to the division."""
    lib = load_library()
    x = _edm_plain(x, "x")
    B = x.shape[0]
    s = torch.empty(B, device=x.device, dtype=torch.float32) if out is None else _edm_plain(out, "out")
    per_image = x[0].numel()
    nbytes = lib.danra_edm_dynamic_threshold_workspace(B, ctypes.c_size_t(per_image))
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    with torch.cuda.device(x.device):
        _check(lib, lib.danra_edm_dynamic_threshold(_ptr(x), _ptr(s), B, ctypes.c_size_t(per_image), percentile, _ptr(ws),
                                                    _stream()), "danra_edm_dynamic_threshold")
    return s


def edm_euler(images_hat, net_out, model_out, thresholds, d, images_next, net_in_next, c_skip, c_out, clamp, sigma_hat, dt,
              c_in_next):
    lib = load_library()
    B = images_hat.shape[0]
    with torch.cuda.device(images_hat.device):
        _check(lib, lib.danra_edm_euler(
            _ptr(_edm_plain(images_hat, "images_hat")), _ptr(_edm_plain(net_out, "net_out")),
            _ptr(_edm_plain(model_out, "model_out")), _ptr(_edm_plain(thresholds, "thresholds")), _ptr(_edm_plain(d, "d")),
            _ptr(_edm_plain(images_next, "images_next")), _ptr(_edm_plain(net_in_next, "net_in_next")), B,
            ctypes.c_size_t(images_hat[0].numel()), c_skip, c_out, int(bool(clamp)), sigma_hat, dt, c_in_next, _stream()),
            "danra_edm_euler")


def edm_heun(images_hat, images_next, net_out, model_out, thresholds, d, images_out, c_skip, c_out, clamp, sigma_next,
             half_dt):
    lib = load_library()
    B = images_hat.shape[0]
    with torch.cuda.device(images_hat.device):
        _check(lib, lib.danra_edm_heun(
            _ptr(_edm_plain(images_hat, "images_hat")), _ptr(_edm_plain(images_next, "images_next")),
            _ptr(_edm_plain(net_out, "net_out")), _ptr(_edm_plain(model_out, "model_out")),
            _ptr(_edm_plain(thresholds, "thresholds")), _ptr(_edm_plain(d, "d")), _ptr(_edm_plain(images_out, "images_out")),
            B, ctypes.c_size_t(images_hat[0].numel()), c_skip, c_out, int(bool(clamp)), sigma_next, half_dt, _stream()),
            "danra_edm_heun")


def edm_renoise(images, scale, noise=None, seed=0, offset=0):
    lib = load_library()
    with torch.cuda.device(images.device):
        _check(lib, lib.danra_edm_renoise(_ptr(_edm_plain(images, "images")), _ptr(_edm_plain(noise, "noise")),
                                          ctypes.c_size_t(images.numel()), scale,
                                          NOISE_PHILOX if noise is None else NOISE_HOST, _u64(seed), _u64(offset), _stream()),
               "danra_edm_renoise")


def edm_finish(images, unnormalize, inpaint=None, mask=None):
    lib = load_library()
    B, C, HW = _edm_dims(_edm_plain(images, "images"))
    with torch.cuda.device(images.device):
        _check(lib, lib.danra_edm_finish(_ptr(images), _ptr(_edm_plain(inpaint, "inpaint")), _ptr(_edm_mask(mask, images)),
                                         B, C, HW, int(bool(unnormalize)), _stream()), "danra_edm_finish")
