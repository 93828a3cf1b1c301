"""Attention-UNet of the "clean application", with the call surface of the reference's
``DDPM_clean_application/src/unet.py`` (same class names, constructor signatures, parameter names and
registration order, so its drivers and checkpoints work unchanged).

It is the network of ``modules_DANRA_conditional.py`` with three differences (reference unet.py):

* ``ImageSelfAttention`` names its MultiheadAttention ``mha`` and ends with a feed-forward block,
  ``x + Linear(GELU(Linear(LayerNorm(x))))`` (:87-96, :110-119);
* ``Encoder`` takes ``cond_on_lsm`` / ``cond_on_topo`` flags instead of the fields themselves, registers no buffers,
  and concatenates whichever of lsm_cond / topo_cond / cond_img is passed to ``forward`` (:137-168, :232-242);
* ``DiffusionNet(encoder, decoder)`` takes nothing else (:580-595).

Execution paths are those of ``modules_DANRA_conditional``: eval-mode CUDA fp32 calls without gradients run in
``libdanra_hip.so`` (no fallback), everything else is the differentiable torch composition.
"""
from typing import Iterable, Optional

import torch
import torch.nn as nn

from . import modules_DANRA_conditional as _base
from .modules_DANRA_conditional import FMAP_CHANNELS, SinusoidalEmbedding, _time_projections
from .resnet import BasicBlock, ResNet

__all__ = ["SinusoidalEmbedding", "ImageSelfAttention", "Encoder", "DecoderBlock", "Decoder", "DiffusionNet",
           "ResNet", "BasicBlock", "torch", "nn", "Optional", "Iterable"]


class ImageSelfAttention(nn.Module):
    """LayerNorm -> multi-head self-attention -> + input -> + feed-forward (reference unet.py:66-119)."""

    def __init__(self, input_channels: int, n_heads: int):
        super().__init__()
        self.input_channels = input_channels
        self.n_heads = n_heads
        self.mha = nn.MultiheadAttention(self.input_channels, self.n_heads, batch_first=True)
        self.layernorm = nn.LayerNorm([self.input_channels])
        self.ff = nn.Sequential(nn.LayerNorm([self.input_channels]),
                                nn.Linear(self.input_channels, self.input_channels),
                                nn.GELU(),
                                nn.Linear(self.input_channels, self.input_channels))

    def forward(self, x: torch.Tensor):
        _, C, H, W = x.shape
        tokens = x.reshape(-1, C, H * W).permute(0, 2, 1)
        normed = self.layernorm(tokens)
        attended = _base.self_attention(self.mha, normed) + tokens
        attended = attended + self.ff(attended)
        return attended.permute(0, 2, 1).reshape(-1, C, H, W)


class Encoder(ResNet):
    """ResNet-18 encoder with a time bias and self-attention after every stage (reference unet.py:123-348)."""

    def __init__(self, input_channels: int, time_embedding: int, block=BasicBlock, block_layers: list = [2, 2, 2, 2],
                 n_heads: int = 4, num_classes: int = None, cond_on_lsm=True, cond_on_topo=True, cond_on_img=False,
                 cond_img_dim=None):
        self.block = block
        self.block_layers = block_layers
        self.time_embedding = time_embedding
        self.input_channels = input_channels
        self.n_heads = n_heads
        self.num_classes = num_classes
        super().__init__(self.block, self.block_layers)

        # one extra conv1 input channel per conditioning field (reference unet.py:163-168)
        if cond_on_lsm:
            self.input_channels += 1
        if cond_on_topo:
            self.input_channels += 1
        if cond_on_img:
            self.input_channels += cond_img_dim[0]
        self.cond_on_lsm, self.cond_on_topo, self.cond_on_img = bool(cond_on_lsm), bool(cond_on_topo), bool(cond_on_img)

        self.sinusiodal_embedding = SinusoidalEmbedding(self.time_embedding)
        self.time_projection_layers = self.make_time_projections(FMAP_CHANNELS)
        self.attention_layers = self.make_attention_layers(FMAP_CHANNELS)
        self.conv1 = nn.Conv2d(self.input_channels, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, time_embedding)
        del self.maxpool, self.fc, self.avgpool

    pos_encoding = _base.Encoder.pos_encoding

    def make_time_projections(self, fmap_channels: Iterable[int]):
        return _time_projections(self.time_embedding, fmap_channels)

    def make_attention_layers(self, fmap_channels: Iterable[int]):
        return nn.ModuleList([ImageSelfAttention(ch, self.n_heads) for ch in fmap_channels])

    def input_planes(self, x, cond_img, lsm_cond, topo_cond):
        """Channel order x, lsm, topo, cond; each conditioning tensor is concatenated iff it is passed
        (reference unet.py:232-242)."""
        return [x] + [p for p in (lsm_cond, topo_cond, cond_img) if p is not None]

    forward = _base.Encoder.forward


class DecoderBlock(_base.DecoderBlock):
    """convT(k2,s2) - IN - conv3x3 - IN - (+skip) - (+time bias, attention) - activation
    (reference unet.py:353-464)."""

    attention_class = ImageSelfAttention

    @staticmethod
    def shape_mismatch_message(prev_fmap, out):
        return 'feature maps must be of same shape. Shape of prev_fmap: {}, shape of output: {}'.format(
            prev_fmap.shape, out.shape)


class Decoder(_base.Decoder):
    """Four up-sampling blocks with additive skips and a final 64 -> output_channels block
    (reference unet.py:469-573)."""

    block_class = DecoderBlock


class DiffusionNet(_base.DiffusionNet):
    """Encoder + decoder (reference unet.py:575-616); dispatches to the HIP path when eligible."""

    encoder_class = Encoder
    decoder_class = Decoder

    def __init__(self, encoder: Encoder, decoder: Decoder):
        super().__init__(encoder, decoder)

    def hip_geometry(self):
        e = self.encoder
        return dict(has_lsm=e.cond_on_lsm, has_topo=e.cond_on_topo, has_cond=e.cond_on_img, attn_ff=True)

    def hip_call_arguments(self, y, cond_img, lsm_cond, topo_cond):
        e = self.encoder
        passed = 1 + sum(int(p.shape[1]) for p in (lsm_cond, topo_cond, cond_img) if p is not None)
        if passed != e.conv1.in_channels:  # the reference fails the same call inside conv1
            raise RuntimeError(f"conv1 expects {e.conv1.in_channels} input channels, but the call concatenates {passed} "
                               "(x plus the conditioning tensors that were passed)")
        for flag, plane, name in ((e.cond_on_lsm, lsm_cond, 'lsm_cond'), (e.cond_on_topo, topo_cond, 'topo_cond'),
                                  (e.cond_on_img, cond_img, 'cond_img')):
            if flag != (plane is not None):
                raise RuntimeError(f"the native path takes {name} exactly when the encoder was built for it")
        return y, cond_img, lsm_cond, topo_cond
