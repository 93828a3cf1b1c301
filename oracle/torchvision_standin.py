"""Stand-in for ``torchvision.models.resnet.{ResNet, BasicBlock}``.

TEST INFRASTRUCTURE ONLY.  torchvision is not installed in this image, and the
reference network subclasses its ``ResNet`` (reference
DDPM_DANRA_conditional/modules_DANRA_conditional.py:14,117,154).  torchvision's
source is not part of the reference tree (the reference pins torchvision=0.16.1,
DDPM_DANRA_conditional_lumi/torch_env.yml:270), so this file restates the published
algorithm of that version for the two classes the reference uses:

* ``BasicBlock(inplanes, planes, stride, downsample)``:
  conv3x3(stride) -> BN -> ReLU -> conv3x3 -> BN -> (+ downsample(x) or x) -> ReLU,
  ``expansion = 1``, convolutions without bias.
* ``ResNet(block, layers)``: creates ``conv1 (3->64, 7x7, s2, p3, no bias), bn1, relu,
  maxpool, layer1..layer4, avgpool, fc`` in that order; ``_make_layer`` adds a
  ``Sequential(conv1x1(stride), BN)`` shortcut whenever the stride is not 1 or the
  channel count changes; kaiming-normal(fan_out, relu) conv init, BN weight 1 / bias 0.

``install()`` registers the stand-in under ``sys.modules['torchvision(.models(.resnet))']``
so that the unmodified reference files import.  Nothing outside ``oracle/`` and
``tests/`` may import this module.
"""
import sys
import types

import torch
import torch.nn as nn


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                 base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = self.bn1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.bn2(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        out = self.relu(out)
        return out


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None,
                 norm_layer=None):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                conv1x1(self.inplanes, planes * block.expansion, stride),
                norm_layer(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, self.groups,
                        self.base_width, self.dilation, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups,
                                base_width=self.base_width, dilation=self.dilation,
                                norm_layer=norm_layer))
        return nn.Sequential(*layers)


def install():
    """Register the stand-in modules so ``from torchvision.models.resnet import ...`` works."""
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_danra_standin", False):
        return  # a real torchvision is present: use it
    tv = types.ModuleType("torchvision")
    tv._danra_standin = True
    models = types.ModuleType("torchvision.models")
    resnet = types.ModuleType("torchvision.models.resnet")
    resnet.ResNet = ResNet
    resnet.BasicBlock = BasicBlock
    models.resnet = resnet
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.models.resnet"] = resnet
