"""DANRA attention-UNet with the call surface of the reference's ``modules_DANRA_conditional.py``.

Same class names, constructor signatures, parameter/buffer names and registration order as the
reference (DDPM_DANRA_conditional/modules_DANRA_conditional.py:17-616), so its drivers, training
pipelines and checkpoints work unchanged (``from modules_DANRA_conditional import *``).

Two execution paths:

* **HIP path** - ``DiffusionNet.forward`` on CUDA fp32 tensors in eval mode with gradients disabled
  (exactly the situation inside ``DiffusionUtils.sample``) runs entirely in ``libdanra_hip.so``
  (hand-written gfx950 kernels behind the C ABI of ``include/danra_hip.h``).  If the library is
  missing this path raises.  A network geometry the library does not take (``hip_supported``) or an input of another dtype or shape
  is announced with a warning once and evaluated by the torch composition below.
* **autograd path** - training mode, gradients enabled, or CPU tensors: a differentiable
  composition of torch ops with the reference's arithmetic, which is what the reference's training
  pipeline (training_DANRA_conditional.py:775-1016) needs for backward.
"""
from typing import Iterable, Optional

import torch
import torch.nn as nn

from .resnet import BasicBlock, ResNet, conv2d, norm_relu

__all__ = ["SinusoidalEmbedding", "ImageSelfAttention", "Encoder", "DecoderBlock", "Decoder", "DiffusionNet",
           "ResNet", "BasicBlock", "torch", "nn", "Optional", "Iterable"]

FMAP_CHANNELS = [64, 64, 128, 256, 512]  # reference modules:170


class SinusoidalEmbedding(nn.Module):
    """Interleaved sin/cos embedding, base ``n`` (reference modules:17-63).

    The reference fills the table with a Python loop over batch x dim/2 scalar tensor ops; this is the
    same arithmetic vectorised: the integer timestep as fp32 divided by ``n**(2i/dim)`` rounded to fp32."""

    def __init__(self, dim_size, n: int = 10000):
        assert dim_size % 2 == 0, 'dim_size must be even'
        super().__init__()
        self.dim_size = dim_size
        self.n = n

    def divisors(self, device):
        """n**(2i/dim) rounded to fp32, evaluated on the host as the reference does and kept per device (a host-to-device
        copy per call would also be illegal while a training step is being captured into a graph)."""
        cache = self.__dict__.setdefault('_div_cache', {})
        key = (str(device), self.dim_size, self.n)
        if key not in cache:
            cache[key] = torch.tensor([self.n ** (2 * i / self.dim_size) for i in range(self.dim_size // 2)],
                                      dtype=torch.float64).to(device=device, dtype=torch.float32)
        return cache[key]

    def forward(self, x: torch.Tensor):
        div = self.divisors(x.device)
        arg = x.reshape(-1).to(torch.float32)[:, None] / div[None, :]
        return torch.stack((torch.sin(arg), torch.cos(arg)), dim=-1).reshape(arg.shape[0], self.dim_size)


def self_attention(mha: nn.MultiheadAttention, normed: torch.Tensor):
    """``mha(normed, normed, normed, need_weights=False)[0]`` on the differentiable path.  On a CUDA tensor the two
    projections run in ``libdanra_hip.so`` (the library is then mandatory), and where the heads are 16 wide and the token
    count is a multiple of 128 (the 64-channel levels, where the L x L matrices of torch's composition dominate a training
    step's time and memory) so does the scaled-dot-product core, forward and backward; the other levels keep torch's fused
    core between the native projections.  Everything else is torch's own module."""
    if normed.is_cuda and mha.dropout == 0.0 and mha.batch_first and mha._qkv_same_embed_dim:
        from . import hip_engine
        if hip_engine.attention_core_supported(normed, mha.num_heads) or (
                torch.is_grad_enabled() and hip_engine.differentiable_attention_supported(mha, normed)):
            return hip_engine.differentiable_self_attention(mha, normed)
    return mha(normed, normed, normed, need_weights=False)[0]


class ImageSelfAttention(nn.Module):
    """LayerNorm -> multi-head self-attention over the H*W tokens -> + un-normalised input
    (reference modules:67-110)."""

    def __init__(self, input_channels: int, n_heads: int):
        super().__init__()
        self.input_channels = input_channels
        self.n_heads = n_heads
        self.layernorm = nn.LayerNorm(self.input_channels)
        self.attention = nn.MultiheadAttention(self.input_channels, self.n_heads, batch_first=True)

    def forward(self, x: torch.Tensor):
        _, C, H, W = x.shape
        if x.is_cuda:  # training step: entry (map -> normalised tokens) and exit (tokens + residual -> map) as one kernel each
            from . import hip_engine
            if hip_engine.attention_block_supported(self.layernorm, x):
                normed = hip_engine.ln_tokens(self.layernorm, x)
                return hip_engine.tokens_residual_nchw(self_attention(self.attention, normed), x)
        tokens = x.reshape(-1, C, H * W).permute(0, 2, 1)
        normed = self.layernorm(tokens)
        attended = self_attention(self.attention, normed)
        return (attended + tokens).permute(0, 2, 1).reshape(-1, C, H, W)


def _time_projections(time_embedding, channels):
    return nn.ModuleList([nn.Sequential(nn.SiLU(), nn.Linear(time_embedding, ch)) for ch in channels])


def _shared_activation(projections, emb):
    """``SiLU(emb)`` evaluated once for several ``Sequential(SiLU, Linear)`` time projections that all read the same
    embedding (each then applies only its Linear: the same arithmetic with one activation kernel instead of one per
    level); None when a projection is built differently or carries forward hooks, and the caller runs the modules whole."""
    for p in projections:
        if not (type(p) is nn.Sequential and len(p) == 2 and type(p[0]) is nn.SiLU and not p[0].inplace
                and not p._forward_hooks and not p._forward_pre_hooks and not p[0]._forward_hooks
                and not p[0]._forward_pre_hooks):
            return None
    return nn.functional.silu(emb)


class Encoder(ResNet):
    """ResNet-18 encoder with a time bias and self-attention after every stage (reference modules:117-345)."""

    def __init__(self, input_channels: int, time_embedding: int, block=BasicBlock, block_layers: list = [2, 2, 2, 2],
                 n_heads: int = 4, num_classes: int = None, lsm_tensor=None, topo_tensor=None, cond_on_img=False,
                 cond_img_dim=None):
        self.block = block
        self.block_layers = block_layers
        self.time_embedding = time_embedding
        self.input_channels = input_channels
        self.n_heads = n_heads
        self.num_classes = num_classes
        super().__init__(self.block, self.block_layers)

        # one extra input plane per registered conditioning field (reference modules:157-164)
        if lsm_tensor is not None:
            self.register_buffer('lsm', lsm_tensor)
            self.input_channels += 1
        if topo_tensor is not None:
            self.register_buffer('elevation', topo_tensor)
            self.input_channels += 1
        if cond_on_img:
            self.input_channels += cond_img_dim[0]
        self.cond_on_img = bool(cond_on_img)

        self.sinusiodal_embedding = SinusoidalEmbedding(self.time_embedding)
        self.time_projection_layers = self.make_time_projections(FMAP_CHANNELS)
        self.attention_layers = self.make_attention_layers(FMAP_CHANNELS)
        self.conv1 = nn.Conv2d(self.input_channels, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, time_embedding)
        del self.maxpool, self.fc, self.avgpool

    def pos_encoding(self, t, channels):
        """[sin | cos] embedding, base 1000 (reference modules:203-211)."""
        cache = self.__dict__.setdefault('_inv_freq_cache', {})  # evaluated on the host (reference rounding), kept per device
        key = (str(t.device), channels)
        if key not in cache:
            cache[key] = (1.0 / (1000 ** (torch.arange(0, channels, 2).float() / channels))).to(t.device)
        arg = t.repeat(1, channels // 2) * cache[key]
        return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)

    def make_time_projections(self, fmap_channels: Iterable[int]):
        return _time_projections(self.time_embedding, fmap_channels)

    def make_attention_layers(self, fmap_channels: Iterable[int]):
        return nn.ModuleList([ImageSelfAttention(ch, self.n_heads) for ch in fmap_channels])

    def input_planes(self, x, cond_img, lsm_cond, topo_cond):
        """The tensors concatenated in front of conv1, channel order x, lsm, topo, cond (reference modules:228-238):
        lsm/topo whenever the buffers were registered, cond_img whenever it is passed."""
        planes = [x]
        if hasattr(self, 'lsm'):
            planes.append(lsm_cond)
        if hasattr(self, 'elevation'):
            planes.append(topo_cond)
        if cond_img is not None:
            planes.append(cond_img)
        return planes

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: Optional[torch.Tensor] = None,
                cond_img: Optional[torch.Tensor] = None, lsm_cond: Optional[torch.Tensor] = None,
                topo_cond: Optional[torch.Tensor] = None):
        planes = self.input_planes(x, cond_img, lsm_cond, topo_cond)
        x = torch.cat(planes, dim=1) if len(planes) > 1 else x

        emb = self.pos_encoding(t.unsqueeze(-1).type(torch.float), self.time_embedding)
        if y is not None:
            emb = emb + self.label_emb(y)

        act = _shared_activation(self.time_projection_layers, emb)

        def stage(level, fmap):
            proj = self.time_projection_layers[level]
            fmap = fmap + (proj(emb) if act is None else proj[1](act))[:, :, None, None]
            return self.attention_layers[level](fmap)

        fmap1 = stage(0, conv2d(self.conv1, x))   # (the training step's stem convolutions run in the native library)
        h = norm_relu(self.bn1, self.relu, conv2d(self.conv2, fmap1))
        fmap2 = stage(1, self.layer1(h))
        fmap3 = stage(2, self.layer2(fmap2))
        fmap4 = stage(3, self.layer3(fmap3))
        fmap5 = stage(4, self.layer4(fmap4))
        return fmap1, fmap2, fmap3, fmap4, fmap5


class _DenseGradient(torch.autograd.Function):
    """Identity whose backward hands on a contiguous gradient.

    torch's CPU backward of ``instance_norm`` returns wrong values when the incoming gradient is laid out
    [H][W][B][C] (observed with torch 2.10: relative error above 1 against the closed form, while a contiguous gradient
    and the GPU kernel agree with it to 1e-7).  That layout is exactly what a ``batch_first`` MultiheadAttention sends
    back through the token reshape of the attention block, which sits right behind the second InstanceNorm of a
    decoder block, so a CPU training step of this network would silently use wrong gradients."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        return grad if grad.is_cuda or grad.is_contiguous() else grad.contiguous()


class DecoderBlock(nn.Module):
    """convT(k2,s2) - IN - conv3x3 - IN - (+skip) - (+time bias, attention) - activation
    (reference modules:349-460)."""

    attention_class = ImageSelfAttention  # a variant of the network swaps in its own attention block

    def __init__(self, input_channels: int, output_channels: int, time_embedding: int, upsample_scale: int = 2,
                 activation: nn.Module = nn.ReLU, compute_attn: bool = True, n_heads: int = 4):
        super().__init__()
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.upsample_scale = upsample_scale
        self.time_embedding = time_embedding
        self.compute_attn = compute_attn
        self.n_heads = n_heads
        self.attention = self.attention_class(self.output_channels, self.n_heads) if compute_attn else nn.Identity()
        self.sinusiodal_embedding = SinusoidalEmbedding(self.time_embedding)
        self.time_projection_layer = nn.Sequential(nn.SiLU(), nn.Linear(self.time_embedding, self.output_channels))
        self.transpose = nn.ConvTranspose2d(self.input_channels, self.input_channels,
                                            kernel_size=self.upsample_scale, stride=self.upsample_scale)
        self.instance_norm1 = nn.InstanceNorm2d(self.transpose.in_channels)
        self.conv = nn.Conv2d(self.transpose.out_channels, self.output_channels, kernel_size=3, stride=1, padding=1)
        self.instance_norm2 = nn.InstanceNorm2d(self.conv.out_channels)
        self.activation = activation()

    @staticmethod
    def shape_mismatch_message(prev_fmap, out):
        return 'feature maps must be of same shape'

    def forward(self, fmap: torch.Tensor, prev_fmap: Optional[torch.Tensor] = None, t: Optional[torch.Tensor] = None,
                t_act: Optional[torch.Tensor] = None):
        """``t_act``: SiLU of this block's sinusoidal embedding of ``t`` where the caller has formed it already (the
        decoder forms it once for its blocks, whose embeddings are the same parameter-free function)."""
        if fmap.is_cuda:  # up-sampling, the two norms and the convolution run in the native library where it has them
            from .hip_engine import conv2d, conv_transpose2d, instance_norm
            up = conv_transpose2d(self.transpose, fmap)
            out = instance_norm(self.instance_norm2, conv2d(self.conv, instance_norm(self.instance_norm1, up)))
        else:
            out = self.instance_norm2(self.conv(self.instance_norm1(self.transpose(fmap))))
        if out.requires_grad:
            out = _DenseGradient.apply(out)
        if torch.is_tensor(prev_fmap):
            assert (prev_fmap.shape == out.shape), self.shape_mismatch_message(prev_fmap, out)
            out = out + prev_fmap
        if torch.is_tensor(t):
            t_emb = (self.time_projection_layer(self.sinusiodal_embedding(t)) if t_act is None
                     else self.time_projection_layer[1](t_act))
            out = self.attention(out + t_emb[:, :, None, None])
        return self.activation(out)


class Decoder(nn.Module):
    """Four up-sampling blocks with additive skips and a final 64 -> output_channels block
    (reference modules:465-569)."""

    block_class = DecoderBlock

    def __init__(self, last_fmap_channels: int, output_channels: int, time_embedding: int,
                 first_fmap_channels: int = 64, n_heads: int = 4):
        super().__init__()
        self.last_fmap_channels = last_fmap_channels
        self.output_channels = output_channels
        self.time_embedding = time_embedding
        self.first_fmap_channels = first_fmap_channels
        self.n_heads = n_heads
        self.residual_layers = self.make_layers()
        self.final_layer = self.block_class(self.residual_layers[-1].input_channels, self.output_channels,
                                        time_embedding=self.time_embedding, activation=nn.Identity,
                                        compute_attn=False, n_heads=self.n_heads)
        self.final_layer.instance_norm2 = nn.Identity()

    def make_layers(self, n: int = 4):
        layers = []
        in_ch = self.last_fmap_channels
        for i in range(n):
            out_ch = in_ch // 2 if i != (n - 1) else self.first_fmap_channels
            layers.append(self.block_class(in_ch, out_ch, time_embedding=self.time_embedding, compute_attn=True,
                                       n_heads=self.n_heads))
            in_ch = out_ch
        return nn.ModuleList(layers)

    def forward(self, *fmaps, t: Optional[torch.Tensor] = None):
        deepest_first = list(reversed(fmaps))
        out = deepest_first[0]
        t_act = None
        blocks = list(self.residual_layers)
        if torch.is_tensor(t) and blocks:
            first = blocks[0].sinusiodal_embedding
            if all(type(b.sinusiodal_embedding) is SinusoidalEmbedding and not b.sinusiodal_embedding._forward_hooks
                   and (b.sinusiodal_embedding.dim_size, b.sinusiodal_embedding.n) == (first.dim_size, first.n)
                   for b in blocks):
                t_act = _shared_activation([b.time_projection_layer for b in blocks], first(t))
        for idx, block in enumerate(blocks):
            out = block(out, deepest_first[idx + 1], t) if t_act is None else block(out, deepest_first[idx + 1], t, t_act)
        return self.final_layer(out)


class DiffusionNet(nn.Module):
    """Encoder + decoder (reference modules:571-616); dispatches to the HIP path when eligible."""

    def __init__(self, encoder: Encoder, decoder: Decoder, lsm_tensor=None, topo_tensor=None, cond_on_img=False,
                 cond_img_dim=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self._hip = None          # (UNetEngine, device)
        self._hip_signature = None

    # ------------------------------------------------------------------ HIP path plumbing
    encoder_class = Encoder
    decoder_class = Decoder

    def hip_geometry(self):
        """Which conditioning planes conv1 expects and which attention block the network uses."""
        e = self.encoder
        return dict(has_lsm=hasattr(e, 'lsm'), has_topo=hasattr(e, 'elevation'), has_cond=bool(e.cond_on_img),
                    attn_ff=False)

    def hip_call_arguments(self, y, cond_img, lsm_cond, topo_cond):
        """(y, cond_img, lsm_cond, topo_cond) as the native library takes them for this network."""
        e = self.encoder
        if not e.cond_on_img and cond_img is not None:
            self._reject_unexpected_cond()
        return (y if hasattr(e, 'label_emb') else None, cond_img if e.cond_on_img else None,
                lsm_cond if hasattr(e, 'lsm') else None, topo_cond if hasattr(e, 'elevation') else None)

    def hip_supported(self):
        """True when the model has the geometry the native library implements (every reference driver's)."""
        e, d = self.encoder, self.decoder
        if not (isinstance(e, self.encoder_class) and isinstance(d, self.decoder_class)):
            return False
        g = self.hip_geometry()
        return (e.block is BasicBlock
                and len(e.block_layers) == 4 and e.n_heads in (1, 2, 4, 8) and d.n_heads == e.n_heads
                and e.time_embedding == d.time_embedding and e.time_embedding % 8 == 0 and e.time_embedding <= 512
                and d.last_fmap_channels == 512 and d.first_fmap_channels == 64 and 1 <= d.output_channels <= 4
                and e.conv1.in_channels == 1 + int(g['has_lsm']) + int(g['has_topo']) + int(g['has_cond']))

    def _hip_eligible(self, x):
        return (not self.training and not torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 1
                and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and self.hip_supported())

    def hip_engine(self, device):
        """The native engine for `device`, (re)packing weights when any parameter or buffer changed."""
        from .hip_engine import UNetEngine
        device = torch.device(device)
        if self._hip is None or self._hip[1] != device:
            e = self.encoder
            engine = UNetEngine(num_classes=e.num_classes or 0, time_embedding=e.time_embedding,
                                out_channels=self.decoder.output_channels, n_heads=e.n_heads,
                                block_layers=list(e.block_layers), device=device, **self.hip_geometry())
            self._hip = (engine, device)
            self._hip_signature = None
        engine = self._hip[0]
        sd = self.state_dict(keep_vars=True)
        signature = tuple((v.data_ptr(), v._version) for v in sd.values())
        if signature != self._hip_signature:
            engine.pack(sd)
            self._hip_signature = signature
        return engine

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: Optional[torch.Tensor] = None,
                cond_img: Optional[torch.Tensor] = None, lsm_cond: Optional[torch.Tensor] = None,
                topo_cond: Optional[torch.Tensor] = None):
        if self._hip_eligible(x):
            return self.hip_engine(x.device).forward(x, t, *self.hip_call_arguments(y, cond_img, lsm_cond, topo_cond))
        if (not self.training and not torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda
                and not getattr(self, '_warned_not_native', False)):
            # an inference call on the GPU that the native library cannot take is never silent
            import warnings
            self._warned_not_native = True
            why = ('the network geometry is outside what libdanra_hip.so implements' if not self.hip_supported() else
                   f'x has dtype {x.dtype} and shape {tuple(x.shape)} (needs float32 [B, 1, H, W] with H, W multiples of 32)')
            warnings.warn(f'DiffusionNet.forward runs as a torch-op composition, not in the HIP library: {why}',
                          RuntimeWarning, stacklevel=2)
        enc_fmaps = self.encoder(x, t=t, y=y, cond_img=cond_img, lsm_cond=lsm_cond, topo_cond=topo_cond)
        return self.decoder(*enc_fmaps, t=t)

    @staticmethod
    def _reject_unexpected_cond():
        raise RuntimeError("cond_img was passed to a model built with cond_on_img=False "
                           "(conv1 has no input channel for it)")
