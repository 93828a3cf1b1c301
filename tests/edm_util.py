"""Toy networks and draw sources for the tests of the elucidated sampler.  The toy network is made of operations that
round identically on the CPU and on the GPU (products, sums, a cyclic shift), so that a whole sampling run can be
compared bit for bit between the oracle on the CPU and the HIP path on the GPU."""
import torch
from torch import nn


class ToyUnet(nn.Module):
    def __init__(self, lowres_cond=False, self_cond=False):
        super().__init__()
        self.lowres_cond = lowres_cond
        self.self_cond = self_cond
        self.anchor = nn.Parameter(torch.zeros(1))  # gives the module a device
        self.calls = 0

    def forward_with_cond_scale(self, x, noise_cond, *, cond_scale=1., self_cond=None, lowres_cond_img=None,
                                lowres_noise_times=None, **unused):
        self.calls += 1
        out = x * 0.3 - x.roll(1, dims=-1) * 0.2 + noise_cond.reshape(-1, 1, 1, 1) * 0.05
        if self_cond is not None:
            out = out + self_cond * 0.1
        if lowres_cond_img is not None:
            out = out + lowres_cond_img * 0.07 + lowres_noise_times.reshape(-1, 1, 1, 1) * 0.01
        return out * cond_scale


class Tape:
    """Standard normal draws made once on the CPU and handed out in order, to any device."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.draws = []
        self.cursor = 0

    def __call__(self, shape, device=None):
        if self.cursor == len(self.draws):
            self.draws.append(torch.randn(tuple(shape), generator=self.gen))
        z = self.draws[self.cursor]
        assert tuple(z.shape) == tuple(shape)
        self.cursor += 1
        return z if device is None else z.to(device)

    def rewind(self):
        self.cursor = 0
