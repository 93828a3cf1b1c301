"""MI355X-native DDPM sampling + UNet forward path for the DANRA conditional diffusion models.

Python host code mirrors the reference's two hot-path modules; the arithmetic of the inference
path lives in ``libdanra_hip.so`` (sources under ``csrc/``, C ABI in ``include/danra_hip.h``).
"""
from .modules_DANRA_conditional import (BasicBlock, Decoder, DecoderBlock, DiffusionNet, Encoder, ImageSelfAttention,
                                        ResNet, SinusoidalEmbedding)
from .diffusion_DANRA_conditional import DiffusionUtils

__all__ = ["SinusoidalEmbedding", "ImageSelfAttention", "Encoder", "DecoderBlock", "Decoder", "DiffusionNet",
           "DiffusionUtils", "ResNet", "BasicBlock"]
