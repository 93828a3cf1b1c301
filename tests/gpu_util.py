"""Thin test-side wrappers that call the per-op C-ABI entry points of libdanra_hip.so."""
import ctypes

import torch

from diffusionmodelscustom_amd import hip_engine
from diffusionmodelscustom_amd.hip_engine import _check, _ptr, _stream


def to_nhwc(lib, x):
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    _check(lib, lib.danra_nchw_to_nhwc(_ptr(x.contiguous()), _ptr(out), B, C, H * W, _stream()), "nchw_to_nhwc")
    return out


def to_nchw(lib, x):
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    _check(lib, lib.danra_nhwc_to_nchw(_ptr(x.contiguous()), _ptr(out), B, C, H * W, _stream()), "nhwc_to_nchw")
    return out


class Ops:
    """Per-op calls on a packed engine; tensors in/out are NCHW torch tensors on the GPU."""

    def __init__(self, engine):
        self.e = engine
        self.lib = engine.lib
        self.h = engine.handle

    def attention(self, index, x, relu=False):
        B, C, H, W = x.shape
        xin = to_nhwc(self.lib, x)
        out = torch.empty_like(xin)
        _check(self.lib, self.lib.danra_attention(self.h, index, _ptr(xin), _ptr(out), B, H * W, int(relu), _stream()),
               "danra_attention")
        return to_nchw(self.lib, out)

    def conv1(self, planes):
        B, _, H, W = planes[0].shape
        ps = [p.contiguous() for p in planes] + [None] * (4 - len(planes))
        out = torch.empty((B, H // 2, W // 2, 64), device=planes[0].device, dtype=torch.float32)
        _check(self.lib, self.lib.danra_conv1(self.h, _ptr(ps[0]), _ptr(ps[1]), _ptr(ps[2]), _ptr(ps[3]), _ptr(out),
                                              B, H, W, _stream()), "danra_conv1")
        return to_nchw(self.lib, out)

    def conv2_bn_relu(self, x):
        B, C, H, W = x.shape
        xin = to_nhwc(self.lib, x)
        out = torch.empty((B, H // 2, W // 2, 64), device=x.device, dtype=torch.float32)
        _check(self.lib, self.lib.danra_conv2_bn_relu(self.h, _ptr(xin), _ptr(out), B, H, W, _stream()), "danra_conv2")
        return to_nchw(self.lib, out)

    def resnet_layer(self, level, x):
        B, C, H, W = x.shape
        xin = to_nhwc(self.lib, x)
        s = 1 if level == 1 else 2
        cout = 64 << (level - 1)
        out = torch.empty((B, H // s, W // s, cout), device=x.device, dtype=torch.float32)
        _check(self.lib, self.lib.danra_resnet_layer(self.h, level, _ptr(xin), _ptr(out), B, H, W, _stream()),
               "danra_resnet_layer")
        return to_nchw(self.lib, out)

    def decoder_block(self, block, fmap, skip=None, t=None):
        B, C, H, W = fmap.shape
        fin = to_nhwc(self.lib, fmap)
        if block == 4:
            out = torch.empty((B, self.e.out_channels, 2 * H, 2 * W), device=fmap.device, dtype=torch.float32)
            _check(self.lib, self.lib.danra_decoder_block(self.h, 4, _ptr(fin), None, None, _ptr(out), B, H, W,
                                                          _stream()), "danra_decoder_block")
            return out
        sk = to_nhwc(self.lib, skip)
        out = torch.empty_like(sk)
        _check(self.lib, self.lib.danra_decoder_block(self.h, block, _ptr(fin), _ptr(sk), _ptr(t.to(torch.int64)),
                                                      _ptr(out), B, H, W, _stream()), "danra_decoder_block")
        return to_nchw(self.lib, out)

    def time_embed(self, t, y=None):
        B = t.numel()
        dim = self.e.cfg.time_embedding
        width = self.lib.danra_tbias_width(self.h)
        enc = torch.empty((B, dim), device=t.device, dtype=torch.float32)
        dec = torch.empty((B, dim), device=t.device, dtype=torch.float32)
        tb = torch.empty((B, width), device=t.device, dtype=torch.float32)
        _check(self.lib, self.lib.danra_time_embed(self.h, _ptr(t.to(torch.int64)),
                                                   _ptr(None if y is None else y.to(torch.int64)), B, _ptr(enc),
                                                   _ptr(dec), _ptr(tb), _stream()), "danra_time_embed")
        return enc, dec, tb

    def instance_norm(self, x):
        B, C, H, W = x.shape
        xin = to_nhwc(self.lib, x)
        out = torch.empty_like(xin)
        _check(self.lib, self.lib.danra_instance_norm(_ptr(xin), _ptr(out), B, H * W, C, _stream()), "danra_instance_norm")
        return to_nchw(self.lib, out)

    def ddpm_update(self, x, eps, noise, c1, c2, c3):
        x = x.clone()
        _check(self.lib, self.lib.danra_ddpm_update(_ptr(x), _ptr(eps), _ptr(noise), ctypes.c_float(c1),
                                                    ctypes.c_float(c2), ctypes.c_float(c3), x.numel(), _stream()),
               "danra_ddpm_update")
        return x

    def ddpm_update_philox(self, x, eps, c1, c2, c3, seed, step):
        x = x.clone()
        _check(self.lib, self.lib.danra_ddpm_update_philox(_ptr(x), _ptr(eps), ctypes.c_float(c1), ctypes.c_float(c2),
                                                           ctypes.c_float(c3), x.numel(), ctypes.c_uint64(seed),
                                                           ctypes.c_uint64(step), _stream()),
               "danra_ddpm_update_philox")
        return x


def conv3x3(lib, x, w, bias=None, res=None, norm=None, relu=False):
    """danra_conv3x3 on NCHW tensors: x [B, Cin, H, W], w [Cout, Cin, 3, 3] (torch layout), norm = (mean, rstd) [B, Cin]."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xin = to_nhwc(lib, x)
    wp = w.permute(0, 2, 3, 1).contiguous()
    rin = None if res is None else to_nhwc(lib, res)
    out = torch.empty((B, H, W, Cout), device=x.device, dtype=torch.float32)
    mean, rstd = (None, None) if norm is None else (norm[0].contiguous(), norm[1].contiguous())
    _check(lib, lib.danra_conv3x3(_ptr(xin), _ptr(wp), _ptr(bias), _ptr(rin), _ptr(mean), _ptr(rstd), _ptr(out), B, H, W, Cin,
                                  Cout, int(relu), _stream()), "danra_conv3x3")
    return to_nchw(lib, out)


def conv8x8s2(lib, x, w, bias=None, res=None, relu=False):
    """danra_conv8x8s2 on NCHW tensors: x [B, Cin, H, W], w [Cout, Cin, 8, 8] (torch layout) -> [B, Cout, H/2, W/2]."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xin = to_nhwc(lib, x)
    wp = w.permute(0, 2, 3, 1).contiguous()
    rin = None if res is None else to_nhwc(lib, res)
    out = torch.empty((B, H // 2, W // 2, Cout), device=x.device, dtype=torch.float32)
    _check(lib, lib.danra_conv8x8s2(_ptr(xin), _ptr(wp), _ptr(bias), _ptr(rin), _ptr(out), B, H, W, Cin, Cout, int(relu),
                                    _stream()), "danra_conv8x8s2")
    return to_nchw(lib, out)


def conv3x3s2(lib, x, w, bias=None, n_split=0, relu=False):
    """danra_conv3x3s2 on NCHW tensors: x [B, Cin, H, W], w [Cout, Cin, 3, 3] -> [B, Cout, H/2, W/2], or with n_split the pair
    ([B, n_split, ..] with the ReLU, [B, Cout - n_split, ..] without)."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xin = to_nhwc(lib, x)
    wp = w.permute(0, 2, 3, 1).contiguous()
    first = n_split if n_split else Cout
    out = torch.empty((B, H // 2, W // 2, first), device=x.device, dtype=torch.float32)
    out2 = torch.empty((B, H // 2, W // 2, Cout - first), device=x.device, dtype=torch.float32) if n_split else None
    _check(lib, lib.danra_conv3x3s2(_ptr(xin), _ptr(wp), _ptr(bias), _ptr(out), _ptr(out2), B, H, W, Cin, Cout, n_split, int(relu),
                                    _stream()), "danra_conv3x3s2")
    return to_nchw(lib, out) if not n_split else (to_nchw(lib, out), to_nchw(lib, out2))
